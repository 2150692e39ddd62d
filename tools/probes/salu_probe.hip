// Probe (round 3): how many SALU instructions a CU issues per cycle when all its waves run scalar code -- the chunk
// prologue and the staging loop of the march kernel are mostly scalar (box arithmetic in SGPRs, m0 / address set-up per
// LDS-DMA instruction, exec-mask bookkeeping of the step loop), and the scalar unit is shared by the CU's four SIMDs.
// One-wave workgroups, W per CU; each wave runs 8 independent s_add / s_and / s_lshl chains.  Also: a mixed stream of
// one SALU per VALU instruction, to see whether scalar work is hidden behind vector work of the same wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void __launch_bounds__(64) probe(unsigned *out, int iters, unsigned seed)
{
    unsigned s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3, s4 = seed + 4, s5 = seed + 5, s6 = seed + 6, s7 = seed + 7;
    float a0 = (float)threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP16(asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n"
                               "s_add_u32 %4, %4, 3\n s_add_u32 %5, %5, 3\n s_add_u32 %6, %6, 3\n s_add_u32 %7, %7, 3\n"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");)
        } else if (MODE == 1) { // dependent chain: latency
            REP16(asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n"
                               "s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 3\n"
                               : "+s"(s0) :: "scc");)
        } else if (MODE == 2) { // one SALU per VALU (independent)
            REP16(asm volatile("s_add_u32 %0, %0, 3\n v_add_f32 %4, %4, %4\n s_add_u32 %1, %1, 3\n v_add_f32 %5, %5, %5\n"
                               "s_add_u32 %2, %2, 3\n v_add_f32 %6, %6, %6\n s_add_u32 %3, %3, 3\n v_add_f32 %7, %7, %7\n"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "scc");)
        } else if (MODE == 3) { // VALU only, same count as the VALU half of mode 2
            REP16(asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 4) { // v_readlane + a scalar use of it
            REP16(asm volatile("v_readlane_b32 %0, %4, 3\n s_add_u32 %1, %1, %0\n v_readlane_b32 %2, %5, 3\n s_add_u32 %3, %3, %2\n"
                               "v_readlane_b32 %0, %6, 3\n s_add_u32 %1, %1, %0\n v_readlane_b32 %2, %7, 3\n s_add_u32 %3, %3, %2\n"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "scc");)
        } else if (MODE == 5) { // exec-mask bookkeeping as the step loop has it
            REP16(asm volatile("s_and_b64 s[20:21], exec, vcc\n s_or_b64 s[22:23], s[20:21], vcc\n s_andn2_b64 s[24:25], s[22:23], exec\n s_and_saveexec_b64 s[26:27], vcc\n"
                               "s_or_b64 exec, exec, s[26:27]\n s_mov_b64 s[20:21], 0\n s_and_b64 s[22:23], s[24:25], exec\n s_or_b64 s[24:25], s[22:23], s[20:21]\n"
                               ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");)
        } else if (MODE == 6) { // s_nop 0 (what pads the DPP chains)
            REP16(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n");)
        } else if (MODE == 7) { // s_mul_i32 / s_lshl / s_min (the box arithmetic)
            REP16(asm volatile("s_mul_i32 %0, %0, 3\n s_lshl_b32 %1, %1, 1\n s_min_i32 %2, %2, %3\n s_max_i32 %3, %3, %4\n"
                               "s_mul_i32 %4, %4, 5\n s_lshr_b32 %5, %5, 1\n s_sub_u32 %6, %6, %7\n s_addc_u32 %7, %7, 0\n"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + (unsigned)(a0 + a1 + a2 + a3);
}
template <int M> void run(const char *name, int per_wave)
{
    unsigned *out; (void)hipMalloc(&out, 256 * 32 * 64 * 4);
    for (int w : {4, 8, 16}) {
        probe<M><<<256 * w, 64>>>(out, 10, 1u); (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0); probe<M><<<256 * w, 64>>>(out, 2000, 1u); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_cu = 2000.0 * 16 * per_wave * w;
        printf("%-34s waves/CU=%2d  %.2f instructions per CU-cycle (%.2f cycles per instruction per wave) at 2.4 GHz\n", name, w,
               insts_per_cu / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / (2000.0 * 16 * per_wave));
    }
    (void)hipFree(out);
}
int main()
{
    run<0>("s_add_u32 x8 independent", 8);
    run<1>("s_add_u32 dependent chain", 8);
    run<2>("1 s_add : 1 v_add_f32", 8);
    run<3>("v_add_f32 alone (4 per group)", 4);
    run<4>("v_readlane + s_add pairs", 8);
    run<5>("exec-mask bookkeeping", 8);
    run<6>("s_nop 0", 8);
    run<7>("s_mul/lshl/min/max/sub/addc", 8);
    return 0;
}
