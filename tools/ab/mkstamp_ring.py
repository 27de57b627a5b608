"""Patch a scratch copy of k_fused_ring.hip / pvx_api.hip with s_memtime stamps (see buildstamp_ring.sh)."""
import sys
src, dst_k, api_src, dst_api = sys.argv[1:5]
s = open(src).read()
def rep(a, b, cnt=1):
    global s
    assert a in s, a[:70]
    s = s.replace(a, b, cnt)
rep("constexpr int GFR = 8;", '''#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t__; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); __builtin_amdgcn_sched_barrier(0); stacc[i] += t__ - stprev; stprev = t__; } while (0)
constexpr int GFR = 8;''')
rep("    v2f raw[R];\n#pragma unroll\n    for (int r = 0; r < R; r++) raw[r] = pvxc::splat(0.f);\n",
    "    v2f raw[R];\n#pragma unroll\n    for (int r = 0; r < R; r++) raw[r] = pvxc::splat(0.f);\n    unsigned long long stacc[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long stprev = 0;\n    { unsigned long long t0__; asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t0__) :: \"memory\"); stprev = t0__; }\n")
rep("        __builtin_amdgcn_sched_barrier(0);\n        prefetch_part(nsrc, 0);\n", "        __builtin_amdgcn_sched_barrier(0);\n        STAMP(0);\n        prefetch_part(nsrc, 0);\n        STAMP(11);\n")
rep("        dft_regs<R>(z);                                             // stage 1\n", "        dft_regs<R>(z);                                             // stage 1\n        STAMP(1);\n")
rep("        wave_sync();\n        dft_regs<R>(z);                                             // stage 2\n", "        wave_sync();\n        STAMP(2);\n        dft_regs<R>(z);                                             // stage 2\n        STAMP(3);\n")
rep("        wave_sync();\n        // ---- untangle in place", "        wave_sync();\n        STAMP(4);\n        // ---- untangle in place")
rep("        const double lsum = (double)ls0 + (double)ls1;", "        STAMP(5);\n        const double lsum = (double)ls0 + (double)ls1;")
rep("        tot = wave_sum(lsum);\n        wave_sync();\n    };", "        tot = wave_sum(lsum);\n        wave_sync();\n        STAMP(6);\n    };")
rep("        int64_t bn = gb, qn = gq;\n        advance(bn, qn);", "        STAMP(10);\n        int64_t bn = gb, qn = gq;\n        advance(bn, qn);")
rep("        block_sync_lds(dbg_bar);                                           // every slot", "        block_sync_lds(dbg_bar);\n        STAMP(12);                                           // every slot")
rep("        if (flags && it > 0) wait_ge(Ppk + wprev, wid == 0 ? it - 1 : it);\n", "        if (flags && it > 0) wait_ge(Ppk + wprev, wid == 0 ? it - 1 : it);\n        STAMP(14);\n")
rep("            const int nsel = peak_select_block<R, u16>(Ly, Lci, G::CAP, Lsel, K, th, mine, lane);\n", "            const int nsel = peak_select_block<R, u16>(Ly, Lci, G::CAP, Lsel, K, th, mine, lane);\n            STAMP(7);\n")
rep("            if (ng == gs) { flush(ng); ng = 0; }", "            STAMP(8);\n            if (ng == gs) { flush(ng); ng = 0; }\n            STAMP(9);")
rep("        block_sync_lds(dbg_bar);                                    // everyone is done", "        block_sync_lds(dbg_bar);\n        STAMP(13);                                    // everyone is done")
rep("    if (ng > 0) flush(ng);\n}", "    if (ng > 0) flush(ng);\n    if (p.spec_out != nullptr && p.spec_row == -7 && lane == 0) { for (int i = 0; i < 16; i++) atomicAdd((unsigned long long*)p.spec_out + i, stacc[i]); }\n}")
open(dst_k, 'w').write(s)
a = open(api_src).read()
t = "        fp.spec_out = spec_row >= 0 ? p->d_specrow : nullptr; fp.spec_row = spec_row;"
assert t in a
a = a.replace(t, t + "\n        if (getenv(\"PVX_STAMPS\")) { fp.spec_out = p->d_specrow; fp.spec_row = -7; }")
t = "extern \"C\" int pvx_plan_get_fft_mode(const pvx_plan* plan) {"
assert t in a
a = a.replace(t, "extern \"C\" int pvx_debug_stamps(pvx_plan* p, unsigned long long* out, int reset) { if (reset) { (void)hipMemset(p->d_specrow, 0, 16 * 8); return 0; } (void)hipDeviceSynchronize(); (void)hipMemcpy(out, p->d_specrow, 16 * 8, hipMemcpyDeviceToHost); return 0; }\n" + t)
open(dst_api, 'w').write(a)
