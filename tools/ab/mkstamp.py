import sys
src, dst_k, api_src, dst_api = sys.argv[1:5]
s=open(src).read()
def rep(a,b,cnt=1):
    global s
    assert a in s, a[:70]
    s=s.replace(a,b,cnt)
rep("constexpr int GF = 8;",'''#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t__; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); __builtin_amdgcn_sched_barrier(0); stacc[i] += t__ - stprev; stprev = t__; } while (0)
constexpr int GF = 8;''')
rep("    float2* prv = L.bufB;\n","    float2* prv = L.bufB;\n    unsigned long long stacc[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long stprev = 0;\n    { unsigned long long t0__; asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t0__) :: \"memory\"); stprev = t0__; }\n")
rep("        __builtin_amdgcn_sched_barrier(0);\n        prefetch(g + 1, bn, qn);\n        if (g < 0 || q == 0) {",
    "        __builtin_amdgcn_sched_barrier(0);\n        STAMP(0);\n        prefetch(g + 1, bn, qn);\n        STAMP(11);\n        if (g < 0 || q == 0) {")
rep("        dft_regs<R>(z);                                           // stage 1\n", "        dft_regs<R>(z);                                           // stage 1\n        STAMP(1);\n")
rep("        // One prefetch site, after the multiplies, for both kinds of row:", "        STAMP(10);\n        // One prefetch site, after the multiplies, for both kinds of row:")
rep("        wave_sync();\n        dft_regs<R>(z);                                           // stage 2\n","        wave_sync();\n        STAMP(2);\n        dft_regs<R>(z);                                           // stage 2\n        STAMP(3);\n")
rep("        wave_sync();\n        // ---- untangle in place","        wave_sync();\n        STAMP(4);\n        // ---- untangle in place")
rep("        const double lsum = (double)ls0 + (double)ls1;","        STAMP(5);\n        const double lsum = (double)ls0 + (double)ls1;")
rep("            tot = wave_sum(lsum);\n        }\n        wave_sync();\n    };","            tot = wave_sum(lsum);\n        }\n        wave_sync();\n        STAMP(6);\n    };")
rep("            const int nsel = peak_select_block<R>(L.y, L.ci, G::CAP, L.sel, K, th, mine, lane);\n","            const int nsel = peak_select_block<R>(L.y, L.ci, G::CAP, L.sel, K, th, mine, lane);\n            STAMP(7);\n")
rep("            if (ng == G_) { flush(ng); ng = 0; }","            STAMP(8);\n            if (ng == G_) { flush(ng); ng = 0; }\n            STAMP(9);")
rep("    if (ng > 0) flush(ng);\n}","    if (ng > 0) flush(ng);\n    if (p.spec_out != nullptr && p.spec_row == -7 && lane == 0) { for (int i = 0; i < 12; i++) atomicAdd((unsigned long long*)p.spec_out + i, stacc[i]); }\n}")
open(dst_k,'w').write(s)
a=open(api_src).read()
assert "        fp.spec_out = spec_row >= 0 ? p->d_specrow : nullptr; fp.spec_row = spec_row;" in a
a=a.replace("        fp.spec_out = spec_row >= 0 ? p->d_specrow : nullptr; fp.spec_row = spec_row;","        fp.spec_out = spec_row >= 0 ? p->d_specrow : nullptr; fp.spec_row = spec_row;\n        if (getenv(\"PVX_STAMPS\")) { fp.spec_out = p->d_specrow; fp.spec_row = -7; }")
assert "extern \"C\" int pvx_plan_get_fft_mode(const pvx_plan* plan) {" in a
a=a.replace("extern \"C\" int pvx_plan_get_fft_mode(const pvx_plan* plan) {","extern \"C\" int pvx_debug_stamps(pvx_plan* p, unsigned long long* out, int reset) { if (reset) { (void)hipMemset(p->d_specrow, 0, 12 * 8); return 0; } (void)hipDeviceSynchronize(); (void)hipMemcpy(out, p->d_specrow, 12 * 8, hipMemcpyDeviceToHost); return 0; }\nextern \"C\" int pvx_plan_get_fft_mode(const pvx_plan* plan) {")
open(dst_api,'w').write(a)
