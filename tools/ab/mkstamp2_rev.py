"""Patch a scratch copy of k_fused_rev.hip / pvx_api.hip with s_memtime stamps at the section boundaries of the nfft-2048 frame
loop (round-5 sources).  Used by tools/ab/buildstamp2_rev.sh; the tree itself is not touched.  A stamp drains the LDS queue
(s_waitcnt lgkmcnt(0) behind s_memtime), so the stamped build shows where a wave's time goes, not what the real build takes."""
import sys
src, dst_k, api_src, dst_api = sys.argv[1:5]
s = open(src).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, (s.count(a), a[:90])
    s = s.replace(a, b)
NS = 18
rep("constexpr int GFR = 8;", '''#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t__; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); __builtin_amdgcn_sched_barrier(0); stacc[i] += t__ - stprev; stprev = t__; } while (0)
constexpr int GFR = 8;''')
rep("    v2f raw[R];\n#pragma unroll\n    for (int r = 0; r < R; r++) raw[r] = pvxc::splat(0.f);\n",
    "    v2f raw[R];\n#pragma unroll\n    for (int r = 0; r < R; r++) raw[r] = pvxc::splat(0.f);\n    unsigned long long stacc[%d] = {0}; unsigned long long stprev = 0;\n    { unsigned long long t0__; asm volatile(\"s_memtime %%0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t0__) :: \"memory\"); stprev = t0__; }\n" % NS)
# 12: the window's LDS reads have landed (before the multiplies, which wait for the row's samples: vmcnt); 13: samples there
rep("            for (int m = 0; m < R / 2; m++) asm volatile(\"\" : \"+v\"(wq[m]));     // (the reads stay together, above the multiplies)",
    "            for (int m = 0; m < R / 2; m++) asm volatile(\"\" : \"+v\"(wq[m]));\n            STAMP(12); asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); STAMP(13);")
# PVX_STAMP_NOSTORE=1 (blocks_override -13): the flush computes but does not store
rep("            if (valid) {\n                const int oi = nout + __popcll(bal & ((1ull << lnf) - 1ull));", "            if (valid && kargs->blocks_override != -13) {\n                const int oi = nout + __popcll(bal & ((1ull << lnf) - 1ull));")
rep("        if (cnt >= 0) {\n            for (int j = nout + e0; j < K; j += LPF) {", "        if (cnt >= 0 && kargs->blocks_override != -13) {\n            for (int j = nout + e0; j < K; j += LPF) {")
# 14: the very top of the loop body; 15: the row's samples are in their registers (explicit vmcnt(0)), nothing read from LDS yet
rep("        const int qn = gq >= 1 ? gq - 1 : Fi;                       // row g - 1 in its signal (below a zero row: the last row of the signal before)",
    "        STAMP(14);\n        const int qn = gq >= 1 ? gq - 1 : Fi;")
rep("        spectrum(zero_row, nsrc, csrc, maxe, mine, tot);\n", "        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); STAMP(15);\n        spectrum(zero_row, nsrc, csrc, maxe, mine, tot);\n")
# PVX_STAMP_LAT=1 (blocks_override -5): wait for the slide's loads right behind them: 16 = issue, 17 = their raw latency
rep("            nsrc = nullptr;\n        }\n        prefetch_part(nsrc, 0);", "            nsrc = nullptr;\n            if (kargs->blocks_override == -5) { STAMP(16); asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\"); STAMP(17); }\n        }\n        prefetch_part(nsrc, 0);")
# 0: loop top + window multiply
rep("        __builtin_amdgcn_sched_barrier(0);\n        if constexpr (H > 0) {", "        __builtin_amdgcn_sched_barrier(0);\n        STAMP(0);\n        if constexpr (H > 0) {")
# 1..3: the transform's hooks; 4: natural-order store + join twiddles; 5: join + untangle
rep("fft4_quarters(z, dz, t1L, lane, [&]() { prefetch_part(nsrc, 1); }, [&]() { prefetch_part(nsrc, 2); }, [&]() { prefetch_part(nsrc, 3); });",
    "STAMP(1); fft4_quarters(z, dz, t1L, lane, [&]() { STAMP(2); prefetch_part(nsrc, 1); }, [&]() { STAMP(3); prefetch_part(nsrc, 2); }, [&]() { STAMP(4); prefetch_part(nsrc, 3); });")
rep("            wave_sync();\n            join4_untangle<256, F4::RP, 64, IdentityIA, false, false>", "            wave_sync();\n            STAMP(5);\n            join4_untangle<256, F4::RP, 64, IdentityIA, false, false>")
rep("        spectrum(zero_row, nsrc, csrc, maxe, mine, tot);\n", "        spectrum(zero_row, nsrc, csrc, maxe, mine, tot);\n        STAMP(6);\n")
rep("            bool own = false;                                       // every kept peak is remembered by the lane that staged it", "            STAMP(7);\n            bool own = false;")
rep("            wave_sync();\n            if (C <= 64 && p.rad <= 5 && !(th < 0.0 && C < K)) {", "            wave_sync();\n            STAMP(8);\n            if (C <= 64 && p.rad <= 5 && !(th < 0.0 && C < K)) {")
rep("        wave_sync();                                                // cur / Ly / lists are read: free for the row below", "        STAMP(9);\n        wave_sync();                                                // cur / Ly / lists are read: free for the row below")
rep("        gq = qn;\n        if (nsrc != nullptr) { csrc = nsrc; orow = norow; }\n    }", "        gq = qn;\n        if (nsrc != nullptr) { csrc = nsrc; orow = norow; }\n        STAMP(10);\n    }\n    STAMP(11);")
# write-out at the very end of the kernel
rep("        if (f1 > f0) flush(f0, f1);\n    }\n}", "        if (f1 > f0) flush(f0, f1);\n    }\n    { kargs_t qq = kargs; if (qq->spec_out != nullptr && qq->spec_row == -7 && lane == 0) { for (int i = 0; i < %d; i++) atomicAdd((unsigned long long*)qq->spec_out + i, stacc[i]); } }\n}" % NS)
open(dst_k, 'w').write(s)
a = open(api_src).read()
key = "        fp.blocks_override = p->fused_blocks;\n"
assert a.count(key) == 1
a = a.replace(key, key + "        if (getenv(\"PVX_STAMPS\")) { fp.spec_out = p->d_specrow; fp.spec_row = -7; }\n        if (getenv(\"PVX_STAMP_NOSTORE\")) fp.blocks_override = -13;\n        if (getenv(\"PVX_STAMP_LAT\")) fp.blocks_override = -5;\n")
k2 = "extern \"C\" int pvx_plan_get_fft_mode(const pvx_plan* plan) {"
assert a.count(k2) == 1
a = a.replace(k2, "extern \"C\" int pvx_debug_stamps(pvx_plan* p, unsigned long long* out, int reset) { if (reset) { (void)hipMemset(p->d_specrow, 0, %d * 8); return 0; } (void)hipDeviceSynchronize(); (void)hipMemcpy(out, p->d_specrow, %d * 8, hipMemcpyDeviceToHost); return 0; }\n" % (NS, NS) + k2)
open(dst_api, 'w').write(a)
