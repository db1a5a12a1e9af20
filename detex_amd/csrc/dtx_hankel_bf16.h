// Split-bf16 ("bf16x3") engine of the implicit-Hankel contraction: the same
//   Y[k][p] += sum_c sum_j A_c[j][k] * x_c[p + j]
// as dtx_hankel.h, with every float32 operand split into two bfloat16 parts
// (hi = bf16(v), lo = bf16(v - hi)) and the product taken as
//   a*b ~= ahi*bhi + ahi*blo + alo*bhi          (alo*blo ~ 2^-18 relative, dropped)
// on v_mfma_f32_16x16x32_bf16 with float32 accumulation: three bf16 MFMAs at 16x
// the FP32 matrix rate each, i.e. up to 5.3x the FP32-MFMA engine.  For this
// problem the error is that of float32 arithmetic (measured on the oracle's
// emulation: max |dDS| 2e-8 before accumulation round-off), and it is LOCAL to a
// window like the float32 engine's -- there is no transform whose error would
// scale with the largest amplitude in a block.
//
// Operand images
//   A (dense, per K-tile, built at upload): [c][qb = Lp/8][h = hi,lo][KT][8] bf16:
//       one 16-byte slot = 8 consecutive contraction indices of one row.  A step of
//       64 indices is one contiguous 256*KT-byte tile -> staged by LDS-DMA.
//       MFMA A fragment (lane = 16 kg + i): slot (qb = 4 kk + kg, h, row 16 r + i).
//   x (stream): the MFMA B fragment of lane (i, kg) is x_c[s .. s+7] with
//       s = p0 + col + j0 + 32 kk + 8 kg, which is only 2-byte aligned.  The demux
//       kernel therefore writes 8 shifted copies G[h][m][v] = x_c[8 v + m .. 8 v + m + 7]
//       (16-byte slots); lane i always reads copy m = i & 7 at an aligned slot.
//   C/D: identical to the float32 16x16x4 layout (row 4 kg + reg, column i), so the
//       epilogues are shared.
//
// LDS banks (ds_read_b128: 16-lane groups mixing kg and kg+1, 64 dword banks = 16
// slots): A slots of a group are rows i = 0..15 of slot-rows that are multiples of
// KT (= 0 mod 16) apart -> 16 distinct slots, conflict-free.
#pragma once
#include "dtx_hankel.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace dtx {

// column tiles per wave in this engine (B fragments are 4 VGPRs each, so fewer than fp32)
__host__ __device__ constexpr int tn_s(int rm) { return rm >= 9 ? 3 : 4; }
// "wide" variant for the big K-tiles (-DDTX_WIDE=1): 4 waves per block, one per SIMD, with
// the whole 512-entry register file each -> twice the column tiles per wave (same block
// tile), i.e. every A fragment read from LDS feeds twice the MFMAs (-41 % LDS reads).
// Measured (profiles/README.md, "power"): on noise it ties with the 8-wave kernel
// (58.6 vs 59.1 ms) because both sit on the chip's power limit; on all-zero data, where
// nothing throttles, one wave per SIMD hides less latency (51.4 vs 46.7 ms).  Off by default.
__host__ __device__ constexpr int nw_s(int) { return 8; }
__host__ __device__ constexpr int tnw_s(int rm) { return tn_s(rm) * 8 / nw_s(rm); }
// slots per shifted copy in LDS (block tile of 128 * tn columns): NS + 4 == 12 (mod 16)
// makes the B reads conflict-free
__host__ __device__ constexpr int xs_for(int tn) { return (128 * tn + kBQ) / 8 + 4; }
// bytes of LDS one block needs (two buffers of A tile + x image)
__host__ __device__ constexpr int lds_bytes_s(int rm) {
  const int loop = 2 * (256 * 16 * rm + 2 * 8 * xs_for(tn_s(rm)) * 16);
  const int tail = 4 * (2 * 16 * rm + 64);   // srow / doff scratch of the shared epilogue
  return loop > tail ? loop : tail;
}

template <int OFF>
__device__ __forceinline__ i32x4 lds_read128_imm(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_read_b128 offset");
  i32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// g16: shifted bf16 copies of this block's stream planes, [Nc][2][8][V] slots of 16 B,
// already offset to the block's chunk; v0 = p0 / 8 is added inside.
// One bf16 MFMA into an accumulator tile.  The 8-wave kernels let hipcc allocate (VGPR
// accumulators).  The 4-wave "wide" kernels own 256 VGPRs + 256 AGPRs per wave; hipcc
// selects ONE accumulator register class per function and, past 256 accumulator
// registers, copies tiles in and out of AGPRs around every MFMA -- so there the class is
// fixed per tile through the asm constraint: tiles [0, 64) in AGPRs, the rest in VGPRs.
template <int MODE>   // 0: builtin, 1: AGPR tile, 2: VGPR tile
__device__ __forceinline__ void mfma_bf16(f32x4& c, const i32x4& a, const i32x4& b) {
  if constexpr (MODE == 0) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  } else if constexpr (MODE == 1) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  } else {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
}

// NW waves per block, TN column tiles per wave: block tile [16 RM] x [16 TN NW]
template <int RM, int TN, int NW = 8>
__device__ __forceinline__ void hankel_mainloop_bf16x3(f32x4 (&acc)[RM][TN],
                                                       const uint16_t* __restrict__ at16, int Lp,
                                                       const i32x4* __restrict__ g16, int64_t V,
                                                       int64_t v0, int Nc, int jlo, int jhi,
                                                       char* lds) {
  constexpr int KT = 16 * RM;
  constexpr int PT = 16 * TN * NW;
  static_assert(PT % 128 == 0, "block tile is a multiple of 128 columns");
  constexpr int NS = (PT + kBQ) / 8;               // x slots per copy per step
  constexpr int XS = xs_for(PT / 128);             // padded slot stride of a copy in LDS
  constexpr int ABYTES = 256 * KT;                 // A tile of one step
  constexpr int XBYTES = 2 * 8 * XS * 16;
  constexpr int BUF = ABYTES + XBYTES;
  constexpr int UNITS = ABYTES / 1024;             // 1 KiB LDS-DMA units
  constexpr int NGU = (UNITS + NW - 1) / NW;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, i = lane & 15, kg = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int col0 = wave * 16 * TN;
  const int spc = (jhi - jlo) / kBQ;               // steps per channel, [jlo, jhi) in 64s
  const int nsteps = Nc * spc;
  if (nsteps <= 0) return;
  const unsigned lds_base = (unsigned)(unsigned long long)((__attribute__((address_space(3))) char*)lds);

  // x image by LDS-DMA too: the LDS image [16 copies][XS slots] is linear, unit u
  // covers slots 64 u .. 64 u + 63; every lane gathers its slot from the copy row
  // (h*8 + m) of G (pad slots re-read the last real slot)
  constexpr int NXU = 16 * XS / 64;                // 1 KiB units of the x image
  static_assert((16 * XS) % 64 == 0, "x image must be whole LDS-DMA units");
  constexpr int NGX = (NXU + NW - 1) / NW;
  int xsrc[NGX];
#pragma unroll
  for (int m = 0; m < NGX; ++m) {
    const int pos = (wave + NW * m) * 64 + lane;
    const int hm = pos / XS, vl = pos - hm * XS;
    xsrc[m] = (hm < 16) ? hm * 65536 + (vl < NS ? vl : NS - 1) : 0;
  }

#define DTX_S_STAGE(s_, buf_)                                                               \
  do {                                                                                      \
    const int c_ = (s_) / spc;                                                              \
    const int j0_ = jlo + ((s_) - c_ * spc) * kBQ;                                          \
    const char* ab_ = reinterpret_cast<const char*>(at16) +                                 \
                      ((int64_t)c_ * (Lp / 8) + j0_ / 8) * (2 * KT * 16) + lane * 16;        \
    char* ld_ = lds + (buf_) * BUF;                                                         \
    _Pragma("unroll") for (int m = 0; m < NGU; ++m) {                                       \
      const int u_ = wave_u + NW * m;                                                       \
      if ((m + 1) * NW <= UNITS || u_ < UNITS)                                              \
        __builtin_amdgcn_global_load_lds(                                                   \
            (const __attribute__((address_space(1))) void*)(ab_ + u_ * 1024),               \
            (__attribute__((address_space(3))) void*)(ld_ + u_ * 1024), 16, 0, 0);          \
    }                                                                                       \
    const i32x4* gx_ = g16 + (int64_t)c_ * (2 * 8) * V + v0 + j0_ / 8;                      \
    _Pragma("unroll") for (int m = 0; m < NGX; ++m) {                                       \
      const int u_ = wave_u + NW * m;                                                       \
      if ((m + 1) * NW <= NXU || u_ < NXU)                                                 \
        __builtin_amdgcn_global_load_lds(                                                   \
            (const __attribute__((address_space(1))) void*)(gx_ + (int64_t)(xsrc[m] >> 16) * V + \
                                                            (xsrc[m] & 0xFFFF)),            \
            (__attribute__((address_space(3))) void*)(ld_ + ABYTES + u_ * 1024), 16, 0, 0); \
    }                                                                                       \
  } while (0)

  DTX_S_STAGE(0, 0);
  __syncthreads();

  // per-lane fragment bases (bytes inside a buffer)
  const unsigned aoff = (unsigned)((kg * 2 * KT + i) * 16);
  const unsigned boff = (unsigned)(ABYTES + ((i & 7) * XS + col0 / 8 + (i >> 3) + kg) * 16);
  constexpr int NU = 2 * RM;                       // (kk, r) iterations per step

  for (int s = 0; s < nsteps; ++s) {
    const unsigned bufb = lds_base + (unsigned)((s & 1) * BUF);
    const unsigned abase = bufb + aoff, bbase = bufb + boff;
    i32x4 ar[3][2];                                // ring of A fragments: [slot][hi, lo]
    i32x4 br[2][TN][2];                            // B fragments: [kk & 1][t][hi, lo]
    // prologue: B(kk = 0), A(0), A(1)
    static_for<0, TN>([&](auto tc) {
      br[0][tc][0] = lds_read128_imm<(2 * tc) * 16>(bbase);
      br[0][tc][1] = lds_read128_imm<(8 * XS + 2 * tc) * 16>(bbase);
    });
    static_for<0, (NU < 2 ? NU : 2)>([&](auto uc) {
      constexpr int u = uc, kk = u / RM, r = u % RM;
      ar[u % 3][0] = lds_read128_imm<((4 * kk * 2 + 0) * KT + 16 * r) * 16>(abase);
      ar[u % 3][1] = lds_read128_imm<((4 * kk * 2 + 1) * KT + 16 * r) * 16>(abase);
    });
    if (s + 1 < nsteps) DTX_S_STAGE(s + 1, (s + 1) & 1);

    static_for<0, NU>([&](auto uc) {
      constexpr int u = uc, kk = u / RM, r = u % RM;
      // loads younger than A(u): A(u+1) (2) and, if iteration u-1 opened a kk, B(kk+1)
      // (RM == 1: B(kk) is issued one iteration before its use, after A(u): wait for all)
      constexpr int younger =
          RM == 1 ? 0
                  : (u + 1 < NU ? 2 : 0) +
                        ((u >= 1 && (u - 1) % RM == 0 && (u - 1) / RM + 1 < 2) ? 2 * TN : 0);
      lds_wait<younger>();
      static_for<0, TN>([&](auto tc) {
        constexpr int t = tc, md = NW == 8 ? 0 : (r * TN + t < 64 ? 1 : 2);
        mfma_bf16<md>(acc[r][t], ar[u % 3][1], br[kk & 1][t][0]);
      });
      static_for<0, TN>([&](auto tc) {
        constexpr int t = tc, md = NW == 8 ? 0 : (r * TN + t < 64 ? 1 : 2);
        mfma_bf16<md>(acc[r][t], ar[u % 3][0], br[kk & 1][t][1]);
      });
      static_for<0, TN>([&](auto tc) {
        constexpr int t = tc, md = NW == 8 ? 0 : (r * TN + t < 64 ? 1 : 2);
        mfma_bf16<md>(acc[r][t], ar[u % 3][0], br[kk & 1][t][0]);
      });
      // issue: B of the next kk when this iteration opens a kk, then A(u + 2)
      if constexpr (r == 0 && kk + 1 < 2) {
        static_for<0, TN>([&](auto tc) {
          br[(kk + 1) & 1][tc][0] = lds_read128_imm<(2 * tc + 4 * (kk + 1)) * 16>(bbase);
          br[(kk + 1) & 1][tc][1] = lds_read128_imm<(8 * XS + 2 * tc + 4 * (kk + 1)) * 16>(bbase);
        });
      }
      if constexpr (u + 2 < NU) {
        constexpr int u2 = u + 2, kk2 = u2 / RM, r2 = u2 % RM;
        ar[u2 % 3][0] = lds_read128_imm<((4 * kk2 * 2 + 0) * KT + 16 * r2) * 16>(abase);
        ar[u2 % 3][1] = lds_read128_imm<((4 * kk2 * 2 + 1) * KT + 16 * r2) * 16>(abase);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
  }
  // the asm MFMAs are opaque to hipcc's hazard recogniser: let the last ones retire
  // before anything reads the accumulators
  if constexpr (NW != 8) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#undef DTX_S_STAGE
}

}  // namespace dtx
