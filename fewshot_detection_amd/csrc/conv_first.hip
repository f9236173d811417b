// First-layer 3x3 convolution (input with <= 4 channels stored as NHWC4): K = 9 taps x 4 channels = 36 is far too
// short for the LDS-staged implicit-GEMM kernel (two k-chunks per tile: all prologue and epilogue, 2.4 TB/s).  This
// layer is HBM-bound -- it writes 32 channels for every 16-byte input pixel -- so the kernel is organised around the
// stores:  persistent waves, no LDS staging, MFMA operands loaded straight into registers.
//   v_mfma_f32_32x32x2_f32, 18 k-steps; step s, lane half h  <->  tap = s >> 1, channel = 2h + (s & 1):
//                            A[pixel][s,h] = x[pixel + tap][channel]   B[s,h][co] = w[co][channel][tap]
//   so each lane half needs two ADJACENT channels of every tap: one 8-byte load per tap per lane (L1/L2-resident),
//   issued one tile ahead of the MFMAs; 18 + 18 registers of prefetch, 18 of weights (loaded once per wave).
//   D[pixel][co]: 32 lanes store 128 contiguous bytes per pixel.
//   BatchNorm partial sums (sum y, sum y^2 per channel) accumulate in registers over the wave's whole pixel run and
//   leave as ONE partial row per block.
//
// Round 5, split arithmetic (the default fp32 GEMM mode): conv_first_split_kernel below.  The kernel above spends ~600 issued
// instructions per 32-pixel tile around 18 dependent fp32 MFMAs (64 cycles each: 1152 matrix cycles per tile and SIMD) and ran
// at 1.9 TB/s.  The products move to v_mfma_f32_32x32x16_bf16 on three-way split operands (conv.hip: six cross terms, fp32
// accurate): 12 MFMAs of 32 cycles for 3 input channels, operands SWAPPED (rows = output channels, columns = pixels) so that a
// lane owns 4 consecutive channels of one pixel per accumulator quad and stores 16 bytes straight from registers -- no LDS
// patch, no per-row predication (whole 32-pixel tiles inside one image row: W % 32 == 0, every darknet input size).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fsdet.h"
#include "conv_common.hpp"
#include "profile.hpp"
#include "ew_types.hpp"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FirstFwdArgs {
  const float* x; const float* w; const float* bias; void* y; float* partial;     // y: float or bf16 (kernel template)
  unsigned x_ld, y_ld;
  int H, W, cin, Cout;
  long long pixels;
  int ppw;                          // pixels per wave (multiple of 32)
  int wide;                         // y 16-byte aligned, pixel rows whole 16-byte pieces: wide stores through LDS
};

template <typename TO>
__global__ __launch_bounds__(256) void conv_first_kernel(FirstFwdArgs p) {
  __shared__ float s_red[4][32][2];
  __shared__ float s_tile[4][32 * 33];              // one output tile per wave on its way to 16-byte stores
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int co = blockIdx.y * 32 + c;
  float bw[18];
#pragma unroll
  for (int s = 0; s < 18; ++s) {
    const int tap = s >> 1, ci = 2 * h + (s & 1);
    bw[s] = ci < p.cin ? p.w[((size_t)co * p.cin + ci) * 9 + tap] : 0.f;
  }
  const float bv = p.bias ? p.bias[co] : 0.f;
  const long long p_begin = ((long long)blockIdx.x * 4 + wave) * p.ppw;
  const long long p_end = p_begin + p.ppw < p.pixels ? p_begin + p.ppw : p.pixels;
  const char* x_b = reinterpret_cast<const char*>(p.x);
  const bool ch0_ok = 2 * h < p.cin, ch1_ok = 2 * h + 1 < p.cin;       // this half's channel pair inside the true cin
  const unsigned hw = (unsigned)(p.H * p.W);
  float s1 = 0.f, s2 = 0.f;

  auto load = [&](long long base, f32x2 (&pt)[9]) {       // this half's channel pair of the 9 tap pixels of pixel c
    const long long pix = base + c;
    const bool valid = pix < p_end;
    const unsigned upix = (unsigned)(valid ? pix : p_begin);            // < 2^31 (launcher check): 32-bit divisions
    const unsigned rem = upix % hw;
    const int yy = (int)(rem / (unsigned)p.W), xx = (int)(rem - (unsigned)yy * (unsigned)p.W);
    const unsigned off = upix * p.x_ld * 4u + 8u * h;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const bool ok = valid && (unsigned)(yy + dy) < (unsigned)p.H && (unsigned)(xx + dx) < (unsigned)p.W;
      const f32x2 v = *reinterpret_cast<const f32x2*>(x_b + (ok ? off + (unsigned)((dy * p.W + dx) * (int)p.x_ld * 4) : 0u));
      // (padding channels beyond cin carry a zero WEIGHT, but 0 * NaN of an uninitialised pad would still poison the sum:
      // they are masked like the out-of-image taps)
      pt[t] = f32x2{ok && ch0_ok ? v[0] : 0.f, ok && ch1_ok ? v[1] : 0.f};
    }
  };
  auto compute = [&](long long base, const f32x2 (&pt)[9]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pt[s >> 1][s & 1], bw[s], acc, 0, 0, 0);
    }
    // rows of this lane: (r & 3) + 8 * (r >> 2) + 4h.  Straight from the accumulators a lane would store ONE channel of a
    // pixel per instruction (16 four- or two-byte stores: 1.42 GB of output left at 3.2 TB/s, the float4 fill of the same
    // buffer runs at 5.3).  The 32 x 32 tile crosses a wave-private LDS patch instead ([pixel][33 floats]: conflict-free
    // both ways) and leaves as 16-byte pieces: lane l stores channels 4 (l & 7) ... +3 (8 as bf16: lanes 0-3 ... of each
    // pixel group) of pixel (l >> 3) + 8 q.
    char* y_b = reinterpret_cast<char*>(p.y);
    const unsigned ys = p.y_ld * (unsigned)sizeof(TO);
    const int left = (int)(p_end - base) < 32 ? (int)(p_end - base) : 32;       // pixels of this tile inside the wave's run
    if (!p.wide) {                                   // unaligned destinations: one value per lane and instruction
      unsigned off = ((unsigned)base + 4u * h) * ys + (unsigned)co * (unsigned)sizeof(TO);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if ((r & 3) + 8 * (r >> 2) + 4 * h < left) {
          const float v = acc[r];
          fsd_ew::st1<TO>(reinterpret_cast<TO*>(y_b + off), v + bv);
          s1 += v;
          s2 += v * v;
        }
        off += ((r & 3) == 3) ? 5u * ys : ys;
      }
      return;
    }
    float* tile = s_tile[wave];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const float v = acc[r];
      tile[row * 33 + c] = v + bv;
      if (row < left) {
        s1 += v;
        s2 += v * v;
      }
    }
    // (wave-private: the DS unit serves a wave's requests in order, the waitcnt is all the synchronisation needed)
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0)
    if constexpr (sizeof(TO) == 4) {
      const int pc = lane & 7, pr = lane >> 3;          // 8 pieces of 4 channels per pixel, 8 pixels per pass
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = pr + 8 * q;
        if (row < left) {
          const float* t = tile + row * 33 + pc * 4;
          const fsd_ew::f32x4 v = {t[0], t[1], t[2], t[3]};
          *reinterpret_cast<fsd_ew::f32x4*>(y_b + ((unsigned)base + (unsigned)row) * ys + ((unsigned)blockIdx.y * 32u + pc * 4u) * 4u) = v;
        }
      }
    } else {
      const int pc = lane & 3, pr = lane >> 2;          // 4 pieces of 8 channels per pixel, 16 pixels per pass
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = pr + 16 * q;
        if (row < left) {
          const float* t = tile + row * 33 + pc * 8;
          fsd_ew::f32x8 v;
          v.lo = fsd_ew::f32x4{t[0], t[1], t[2], t[3]};
          v.hi = fsd_ew::f32x4{t[4], t[5], t[6], t[7]};
          fsd_ew::st8(reinterpret_cast<fsd_ew::bf16_t*>(y_b + ((unsigned)base + (unsigned)row) * ys + ((unsigned)blockIdx.y * 32u + pc * 8u) * 2u), v);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);               // the patch is read before the next tile overwrites it
  };

  if (p_begin < p_end) {
    f32x2 p0[9], p1[9];
    load(p_begin, p0);
    for (long long base = p_begin; base < p_end; base += 64) {
      load(base + 32, p1);
      __builtin_amdgcn_sched_barrier(0);
      compute(base, p0);
      __builtin_amdgcn_sched_barrier(0);
      load(base + 64, p0);
      __builtin_amdgcn_sched_barrier(0);
      if (base + 32 < p_end) compute(base + 32, p1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (p.partial == nullptr) return;
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  if (h == 0) { s_red[wave][c][0] = s1; s_red[wave][c][1] = s2; }
  __syncthreads();
  if (threadIdx.x < 32) {
    float* dst = p.partial + ((size_t)blockIdx.x * p.Cout + co) * 2;
    dst[0] = s_red[0][c][0] + s_red[1][c][0] + s_red[2][c][0] + s_red[3][c][0];
    dst[1] = s_red[0][c][1] + s_red[1][c][1] + s_red[2][c][1] + s_red[3][c][1];
  }
}


typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// K-slot layout of the split kernel.  A lane half h owns SLOTS = 8 * KS consecutive-in-program-order k-slots (slot q = 8 s + j
// is element j of its operand register in k-step s).  Half 0 carries taps 0..4, half 1 taps 5..8, CIN values per tap, zeros
// behind: (tap, ci) of slot q in half h = (5 h + q / CIN, q % CIN) while that tap is <= (h ? 8 : 4).  A (weights) and B
// (pixels) use the same map, so any order is a valid reduction order.
template <int CIN> struct FirstSlots {
  static constexpr int KS = CIN == 3 ? 2 : 3;       // k-steps of 16: 27 -> 32 slots, 36 -> 48 slots
  static constexpr int SLOTS = 8 * KS;              // per lane half
  static constexpr int TAPS = 5;                    // float4 loads per lane and tile
};

typedef float f32x3_t __attribute__((ext_vector_type(3)));
template <int CIN> struct PixelVec { typedef fsd_conv::f32x4 type; };
// three input channels: a 12-byte load.  With a 16-byte load the compiler knows the fourth component is dead, hands that register
// to address arithmetic right behind the load and has to wait for the load first (WAW): s_waitcnt vmcnt(0) inside the load phase
template <> struct PixelVec<3> { typedef f32x3_t type; };

template <int CIN, typename TO>
__global__ __launch_bounds__(256) void conv_first_split_kernel(FirstFwdArgs p) {
  typedef FirstSlots<CIN> L;
  typedef typename PixelVec<CIN>::type pix_t;
  constexpr int KS = L::KS, SLOTS = L::SLOTS;
  __shared__ float s_red[4][32][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int co0 = blockIdx.y * 32;
  // ---- A operand: the 3x3xCIN weights of output channel co0 + c, three bf16 planes, constant over the wave's run ----
  bf16x8_t wa[3][KS];
  {
    float wv[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
      const int tap = 5 * h + q / CIN, ci = q % CIN;
      const bool ok = q < 5 * CIN && tap < 9 && ci < p.cin;
      wv[q] = ok ? p.w[((size_t)(co0 + c) * p.cin + ci) * 9 + tap] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      uint2 h0, m0, l0, h1, m1, l1;
      fsd_conv::split3(fsd_conv::f32x4{wv[8 * s], wv[8 * s + 1], wv[8 * s + 2], wv[8 * s + 3]}, h0, m0, l0);
      fsd_conv::split3(fsd_conv::f32x4{wv[8 * s + 4], wv[8 * s + 5], wv[8 * s + 6], wv[8 * s + 7]}, h1, m1, l1);
      wa[0][s] = __builtin_bit_cast(bf16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
      wa[1][s] = __builtin_bit_cast(bf16x8_t, make_uint4(m0.x, m0.y, m1.x, m1.y));
      wa[2][s] = __builtin_bit_cast(bf16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
    }
  }
  // bias of the block's 32 channels, read back per accumulator quad in the bf16 epilogue (accumulator r <-> channel
  // 8 (r >> 2) + 4 h + (r & 3)); in LDS, not in 16 registers
  __shared__ __attribute__((aligned(16))) float s_bias[32];
  __shared__ __attribute__((aligned(16))) float s_tile[4 * 32 * 36];       // per wave: [32 px][32 ch + 4] fp32 or [32 px][16 + 4 dwords] bf16
  if (threadIdx.x < 32) s_bias[threadIdx.x] = p.bias ? p.bias[co0 + threadIdx.x] : 0.f;
  __syncthreads();
  // ---- this wave's run of 32-pixel tiles (a tile lies inside one image row) ----
  const long long tiles = p.pixels >> 5;
  const long long t_begin = ((long long)blockIdx.x * 4 + wave) * (long long)(p.ppw >> 5);
  long long t_end = t_begin + (p.ppw >> 5);
  if (t_end > tiles) t_end = tiles;
  // tap geometry of this lane's five loads (tile-invariant)
  int t_dy[5], t_dx[5], t_off[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int tap = 5 * h + i;
    t_dy[i] = tap < 9 ? tap / 3 - 1 : (1 << 20);                 // tap 9 (half 1, fifth load): never inside the image
    t_dx[i] = tap < 9 ? tap % 3 - 1 : 0;
    t_off[i] = tap < 9 ? (t_dy[i] * p.W + t_dx[i]) * (int)p.x_ld * 4 : 0;
  }
  const char* x_b = reinterpret_cast<const char*>(p.x);
  char* y_b = reinterpret_cast<char*>(p.y);
  const unsigned ys = p.y_ld * (unsigned)sizeof(TO);
  // BatchNorm partial sums of the raw accumulators.  fp32 output: taken where the tile LEAVES the LDS patch -- a lane always stores
  // the same four channels (4 (lane & 7) ..), so 4 + 4 accumulators instead of 16 + 16 (3 waves per SIMD instead of 2).  bf16
  // output: the patch holds rounded values, the sums are taken from the accumulators (16 channels per lane).  (Routing the bf16
  // output through an fp32 patch as well -- 8 + 8 sum registers -- came out at 162 registers, 2 waves per SIMD.)
  constexpr int NS = sizeof(TO) == 4 ? 4 : 16;
  float s1[NS], s2[NS];
#pragma unroll
  for (int r = 0; r < NS; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
  fsd_conv::f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (sizeof(TO) == 4 && p.bias) bias4 = *reinterpret_cast<const fsd_conv::f32x4*>(p.bias + co0 + 4 * (lane & 7));

  // position of the tile being LOADED: image row yy, first column x0, pixel index pix0 (wave-uniform)
  const int tiles_per_row = p.W >> 5;
  long long t_load = t_begin;
  int x0 = (int)(t_begin % tiles_per_row) * 32;
  int yy = (int)((t_begin / tiles_per_row) % p.H);
  // The loads are UNCONDITIONAL from clamped (always mapped) addresses and the zero-select happens at use (mask): a select right
  // behind its load makes the compiler wait for the data inside the load phase -- s_waitcnt vmcnt(0) after the first load of every
  // tile, which also waits for the previous tile's stores: no prefetch at all, 12 us per tile and wave (PMC: 73 % of the wave time
  // in s_waitcnt, 3 TB/s; the fp32-MFMA kernel above has the same flaw).
  auto load = [&](pix_t (&v)[5], unsigned& mask) {
    const bool live = t_load < t_end;
    const unsigned off = ((unsigned)(t_load << 5) + (unsigned)c) * p.x_ld * 4u;
    mask = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const bool ok = live && (unsigned)(yy + t_dy[i]) < (unsigned)p.H && (unsigned)(x0 + c + t_dx[i]) < (unsigned)p.W;
      v[i] = *reinterpret_cast<const pix_t*>(x_b + (ok ? off + (unsigned)t_off[i] : 0u));
      mask |= (ok ? 1u : 0u) << i;
    }
    ++t_load;
    x0 += 32;
    if (x0 == p.W) { x0 = 0; if (++yy == p.H) yy = 0; }
  };
  auto compute = [&](long long tile_i, const pix_t (&vr)[5], unsigned mask) {
    pix_t v[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < CIN; ++j) v[i][j] = (mask >> i) & 1u ? vr[i][j] : 0.f;
    // B operand: this lane's SLOTS values of pixel 32 tile_i + c, split into three planes
    bf16x8_t xb[3][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      fsd_conv::f32x4 g0, g1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q0 = 8 * s + j, q1 = 8 * s + 4 + j;
        g0[j] = q0 < 5 * CIN ? v[q0 / CIN][q0 % CIN] : 0.f;
        g1[j] = q1 < 5 * CIN ? v[q1 / CIN][q1 % CIN] : 0.f;
      }
      uint2 h0, m0, l0, h1, m1, l1;
      fsd_conv::split3(g0, h0, m0, l0);
      fsd_conv::split3(g1, h1, m1, l1);
      xb[0][s] = __builtin_bit_cast(bf16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
      xb[1][s] = __builtin_bit_cast(bf16x8_t, make_uint4(m0.x, m0.y, m1.x, m1.y));
      xb[2][s] = __builtin_bit_cast(bf16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
    }
    // the six cross terms down to 2^-16 relative, smallest first (rows = channels: A = weights, columns = pixels: B = input).
    // (Two alternating chains into two accumulators cost 16 registers = a wave per SIMD: 0.47 -> 0.49 ms; the kernel is bound by
    // latency per wave, occupancy is what hides it.)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[TA[t]][s], xb[TB[t]][s], acc, 0, 0, 0);
    // lane (c, h): pixel 32 tile + c, channels co0 + 8 g + 4 h + (0..3) in accumulators 4 g .. 4 g + 3
    if constexpr (sizeof(TO) != 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s1[r] += acc[r];
        s2[r] = __builtin_fmaf(acc[r], acc[r], s2[r]);
      }
    }
    // The tile crosses a wave-private LDS patch so that every store instruction writes WHOLE lines: straight from the
    // accumulators a lane holds four channel quads of one pixel, i.e. a store instruction would touch 32 lines with 32 bytes each
    // (measured: 655 us at B = 64 against 503 for the fp32-MFMA kernel).  Quads go in as they lie (ds_write_b128 / b64, rows
    // padded by 16 bytes: conflict-free), rows come out as 16-byte pieces, 8 (fp32) / 4 (bf16) lanes per pixel row.
    if constexpr (sizeof(TO) == 4) {
      float* tile = s_tile + wave * (32 * 36);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const fsd_conv::f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *reinterpret_cast<fsd_conv::f32x4*>(tile + c * 36 + 8 * g + 4 * h) = o;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): wave-private patch, the DS unit serves a wave in order
      const int pc = lane & 7, pr = lane >> 3;
      const unsigned base = (unsigned)(tile_i << 5) * ys + (unsigned)(co0 + pc * 4) * 4u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rowp = pr + 8 * q;
        const fsd_conv::f32x4 v4 = *reinterpret_cast<const fsd_conv::f32x4*>(tile + rowp * 36 + pc * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s1[j] += v4[j];
          s2[j] = __builtin_fmaf(v4[j], v4[j], s2[j]);
        }
        *reinterpret_cast<fsd_conv::f32x4*>(y_b + base + (unsigned)rowp * ys) = v4 + bias4;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);               // the patch is read before the next tile overwrites it
    } else {
      unsigned* tile = reinterpret_cast<unsigned*>(s_tile) + wave * (32 * 20);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const fsd_conv::f32x4 b4 = *reinterpret_cast<const fsd_conv::f32x4*>(s_bias + 8 * g + 4 * h);
        uint2 pk;
        pk.x = (unsigned)fsd_ew::f32_to_bf16(acc[4 * g] + b4[0]) | ((unsigned)fsd_ew::f32_to_bf16(acc[4 * g + 1] + b4[1]) << 16);
        pk.y = (unsigned)fsd_ew::f32_to_bf16(acc[4 * g + 2] + b4[2]) | ((unsigned)fsd_ew::f32_to_bf16(acc[4 * g + 3] + b4[3]) << 16);
        *reinterpret_cast<uint2*>(tile + c * 20 + 4 * g + 2 * h) = pk;         // channels 8 g + 4 h .. + 3 = dwords 4 g + 2 h, + 1
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      const int pc = lane & 3, pr = lane >> 2;
      const unsigned base = (unsigned)(tile_i << 5) * ys + (unsigned)(co0 + pc * 8) * 2u;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int rowp = pr + 16 * q;
        const uint4 v4 = *reinterpret_cast<const uint4*>(tile + rowp * 20 + pc * 4);
        *reinterpret_cast<uint4*>(y_b + base + (unsigned)rowp * ys) = v4;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
  };

  if (t_begin < t_end) {
    // Two tiles in flight.  (Peeling the first tile so that the loop is entered in its steady state -- the compiler's waitcnt
    // model takes the smaller count of the entry and the back-edge path, so every other tile also waits for the previous tile's
    // stores -- costs 8-30 registers = a wave per SIMD in the bf16 variant: not worth it.)
    pix_t v0[5], v1[5];
    unsigned m0 = 0, m1 = 0;
    load(v0, m0);
    for (long long t = t_begin; t < t_end; t += 2) {
      load(v1, m1);
      __builtin_amdgcn_sched_barrier(0);
      compute(t, v0, m0);
      __builtin_amdgcn_sched_barrier(0);
      load(v0, m0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < t_end) compute(t + 1, v1, m1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (p.partial == nullptr) return;
  if constexpr (sizeof(TO) == 4) {
    // lane l holds channels 4 (l & 7) .. + 3 of the pixel rows l >> 3 (+ 8 q): fold the eight row lanes
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int m = 8; m < 64; m <<= 1) {
        s1[j] += __shfl_xor(s1[j], m, 64);
        s2[j] += __shfl_xor(s2[j], m, 64);
      }
    }
    if (lane < 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s_red[wave][4 * lane + j][0] = s1[j];
        s_red[wave][4 * lane + j][1] = s2[j];
      }
    }
  } else {
    // per-channel sums over the 32 pixel lanes of each half
#pragma unroll
    for (int r = 0; r < NS; ++r) {
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) {
        s1[r] += __shfl_xor(s1[r], m, 64);
        s2[r] += __shfl_xor(s2[r], m, 64);
      }
    }
    if (c == 0) {
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        const int ch = 8 * (r >> 2) + 4 * h + (r & 3);
        s_red[wave][ch][0] = s1[r];
        s_red[wave][ch][1] = s2[r];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    float* dst = p.partial + ((size_t)blockIdx.x * p.Cout + co0 + threadIdx.x) * 2;
    dst[0] = s_red[0][threadIdx.x][0] + s_red[1][threadIdx.x][0] + s_red[2][threadIdx.x][0] + s_red[3][threadIdx.x][0];
    dst[1] = s_red[0][threadIdx.x][1] + s_red[1][threadIdx.x][1] + s_red[2][threadIdx.x][1] + s_red[3][threadIdx.x][1];
  }
}

inline int first_fwd_blocks(long long pixels) {
  long long b = (pixels + 4 * 256 - 1) / (4 * 256);        // at least 256 pixels per wave
  return (int)(b < 1 ? 1 : b > 2048 ? 2048 : b);
}

}  // namespace

extern "C" int fsd_conv3x3_c4_partial_rows(int batch, int height, int width) {
  return first_fwd_blocks((long long)batch * height * width);
}

namespace {

template <typename TO>
int conv_first_impl(const float* x, long long x_ld, const float* w_oihw, const float* bias, TO* y, long long y_ld,
                    float* bn_partial, int batch, int height, int width, int cin, int cout, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !w_oihw || !y || batch < 1 || height < 1 || width < 1) return FSD_ERR_ARG;
  if (cin < 1 || cin > 4 || cout % 32 || x_ld < 4 || (x_ld & 3) || y_ld < cout) return FSD_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) & 15)) return FSD_ERR_ARG;
  const long long pixels = (long long)batch * height * width;
  if ((pixels + width + 66) * (x_ld > y_ld ? x_ld : y_ld) * 4 >= 0xffffffffLL) return FSD_ERR_UNSUPPORTED;   // 32-bit byte offsets
  const int blocks = first_fwd_blocks(pixels);
  FirstFwdArgs a;
  a.x = x; a.w = w_oihw; a.bias = bias; a.y = y; a.partial = bn_partial;
  a.x_ld = (unsigned)x_ld; a.y_ld = (unsigned)y_ld;
  a.H = height; a.W = width; a.cin = cin; a.Cout = cout; a.pixels = pixels;
  const long long per = (pixels + (long long)blocks * 4 - 1) / ((long long)blocks * 4);
  a.ppw = (int)((per + 63) / 64 * 64);
  static const char* wide_env = FSD_TUNE("FSD_FIRST_WIDE");           // tuning aid: 0 = narrow stores
  a.wide = (!(wide_env && wide_env[0] == '0') && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (y_ld * sizeof(TO)) % 16 == 0) ? 1 : 0;
  fsd_prof::Scope prof(fsd_prof::kFirst, (double)batch * height * width * (16.0 + (double)sizeof(TO) * cout), stream);
  // split arithmetic (the default of the fp32 GEMMs, FSD_F32_SPLIT): the bf16-MFMA kernel for whole 32-pixel row tiles and
  // 16-byte stores; everything else (native arithmetic, odd widths, unaligned destinations) keeps the fp32-MFMA kernel
  static const char* v2_env = FSD_TUNE("FSD_FIRST_SPLIT");           // tuning aid: 0 = always the fp32-MFMA kernel
  if (!(v2_env && v2_env[0] == '0') && fsd_conv::f32_split_on() && a.wide && width % 32 == 0) {
    // <3> never reads the fourth component; <4> multiplies every component it loads, so it takes the full pixel only
    // (cin = 1, 2: the fp32-MFMA kernel below, which masks the padding channels -- ADVICE r5)
    if (cin == 3) {
      FSD_LAUNCH((conv_first_split_kernel<3, TO>), dim3(blocks, cout / 32), dim3(256), 0, stream, a);
      return (int)hipGetLastError();
    }
    if (cin == 4) {
      FSD_LAUNCH((conv_first_split_kernel<4, TO>), dim3(blocks, cout / 32), dim3(256), 0, stream, a);
      return (int)hipGetLastError();
    }
  }
  FSD_LAUNCH(conv_first_kernel<TO>, dim3(blocks, cout / 32), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int fsd_conv3x3_c4_fwd(const float* x, long long x_ld, const float* w_oihw, const float* bias, float* y,
                                  long long y_ld, float* bn_partial, int batch, int height, int width, int cin, int cout,
                                  hipStream_t stream) {
  return conv_first_impl<float>(x, x_ld, w_oihw, bias, y, y_ld, bn_partial, batch, height, width, cin, cout, stream);
}

/* bf16 mode: the same fp32 arithmetic (fp32 input pixels, fp32 weights), the raw output stored as bfloat16; the BatchNorm
 * partial sums still come from the fp32 accumulators. */
extern "C" int fsd_conv3x3_c4_fwd_h(const float* x, long long x_ld, const float* w_oihw, const float* bias, void* y_bf16,
                                    long long y_ld, float* bn_partial, int batch, int height, int width, int cin, int cout,
                                    hipStream_t stream) {
  return conv_first_impl<fsd_ew::bf16_t>(x, x_ld, w_oihw, bias, static_cast<fsd_ew::bf16_t*>(y_bf16), y_ld, bn_partial, batch,
                                         height, width, cin, cout, stream);
}
