// Stamping / ablation hooks of the tiled GEMM kernels -- DEVELOPMENT ONLY.
// The product headers (autosmoothquant_amd/csrc/asq_gemm_p8.h and the kernels that share its schedule) define every P8_* / P8H_* / P4_* hook as an empty
// statement; a probe translation unit (tools/ubench/p8_probe.hip, clock_probe.hip) defines ASQ_P8_PROBE before it includes the sources, and only then is this
// file pulled in instead: per-block s_memtime / s_memrealtime stamps, phase timers and the XCC / tile bookkeeping the probes print.
// Nothing in libasq_hip.so is built with it (tests/test_abi_cpu.py pins the product's symbol list).
#pragma once

static __device__ unsigned long long p8_dbg[2][4][8];
// ABL & 128: per-block {start, prologue done, loop done, end} in s_memtime ticks, xcc_id, tile id, {start, end} in s_memrealtime (100 MHz) ticks
static __device__ unsigned long long p8_blk[4096][8];

#define P8_ABL_OK(ABL) true
#define P8_BLK(i) do { if constexpr (ABL & 128) { if (wave == 0 && lane == 0 && blockIdx.x < 4096) p8_blk[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } } while (0)
#define P8_BLK_RT(i) do { if constexpr (ABL & 128) { if (wave == 0 && lane == 0 && blockIdx.x < 4096) p8_blk[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define P8_STAMP(i) do { if constexpr (ABL & 32) st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define P8_ACCUM(ph) do { if constexpr (ABL & 32) { _Pragma("unroll") for (int q_ = 0; q_ < 7; ++q_) tacc[ph][q_] += st[q_ + 1] - st[q_]; } } while (0)
// phase timers of gemm_i8_p8 (ABL & 32): declared at the top of the K loop, dumped by block 0's waves 0 / 4 behind it
#define P8_PROBE_TIMERS() unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tacc[4][7] = {}; (void)st; (void)tacc
#define P8_PROBE_DUMP_TIMERS() do { if constexpr (ABL & 32) { if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) \
        for (int a_ = 0; a_ < 4; ++a_) for (int b_ = 0; b_ < 7; ++b_) p8_dbg[wave >> 2][a_][b_] = tacc[a_][b_]; } } while (0)
// end of a block (ABL & 128): drain, stamp, and (gemm_i8_p8) record where the block ran and which tile it had
#define P8_PROBE_END() do { if constexpr (ABL & 128) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); P8_BLK(3); P8_BLK_RT(7); } } while (0)
#define P8_PROBE_END_WHERE(tile_m, tile_n) do { if constexpr (ABL & 128) { \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); P8_BLK(3); P8_BLK_RT(7); \
        if (wave == 0 && lane == 0 && blockIdx.x < 4096) { \
            unsigned xcc_, hwid_; \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid_)); \
            p8_blk[blockIdx.x][4] = (xcc_ & 7) | ((unsigned long long)hwid_ << 8);  /* bits 0-2 XCC; HW_ID above (cu_id 8-11, sh_id 12, se_id 13-15 of it) */ \
            p8_blk[blockIdx.x][5] = (unsigned)((tile_m) * 65536 + (tile_n)); \
        } } } while (0)
// the LDS-DMA helper of some ablation instantiations evaluates base + k on the VALU: pin it again
#define P8_PROBE_PIN_BASE(p) p = uniform_ptr(p)

// gemm_i8_p8h (template parameter PROBE)
#define P8H_BLK(i) do { if constexpr (PROBE) { if (wave == 0 && lane == 0 && blockIdx.x < 4096) p8_blk[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } } while (0)
#define P8H_PROBE_END(tile_m, tile_n) do { if constexpr (PROBE) { \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); P8H_BLK(3); \
        if (wave == 0 && lane == 0 && blockIdx.x < 4096) { \
            unsigned xcc_; \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); \
            p8_blk[blockIdx.x][4] = xcc_; \
            p8_blk[blockIdx.x][5] = (unsigned)((tile_m) * 65536 + (tile_n)); \
        } } } while (0)

// gemm_i8_p4 / gemm_i8_p4x16 (template parameter PROBE)
#define P4_BLK(i) do { if constexpr (PROBE) { if (wave == 0 && lane == 0 && blockIdx.x < 4096) p8_blk[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } } while (0)
#define P4_BLK_RT(i) do { if constexpr (PROBE) { if (wave == 0 && lane == 0 && blockIdx.x < 4096) p8_blk[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define P4_PROBE_END() do { if constexpr (PROBE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); P4_BLK(3); P4_BLK_RT(7); } } while (0)
