// bf16 compute mode of the implicit-GEMM convolution (BASELINE configs C3 / C5: "bf16 in, fp32
// accumulate"): v_mfma_f32_32x32x16_bf16, 16x the fp32 matrix rate.
//
// Activations stay fp32 NHWC in HBM (BatchNorm statistics, activations, loss and master weights are
// fp32); they are rounded to bf16 (v_cvt_pk_bf16_f32, round-to-nearest-even) while being staged into
// LDS.  Weights are pre-packed as bf16, K-major, like the fp32 path.  A k-chunk is 64 elements:
//   A: each thread loads 16 B = 4 fp32 channels of one tap, 8 (BM=128) rows -> 8-byte LDS stores
//   B: each thread loads 16 B = 8 bf16 weights, 4 rows                      -> 16-byte LDS stores
// LDS rows are 64 bf16 = 128 B padded to 144 B, so the ds_read_b128 fragment reads (lane = row,
// 8 consecutive k per lane) are conflict-free exactly as in the fp32 kernel.  One b128 read per operand
// tile feeds one MFMA (K = 16).  With the matrix work 16x cheaper the kernel is bound by tile staging
// (L2 -> LDS), hence the larger 128x128 tile (half the staged bytes per FLOP of a 64x64 tile).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fsdet.h"
#include "conv_common.hpp"
#include "profile.hpp"

namespace {

using namespace fsd_conv;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBKh = 64;     // k-chunk, in bf16 elements
constexpr int kLdh = 72;     // padded LDS row stride, in bf16 elements (144 B)
constexpr int kNT = 256;

__device__ __forceinline__ uint2 pack4(f32x4 v) {
  const __bf16 b0 = (__bf16)v[0], b1 = (__bf16)v[1], b2 = (__bf16)v[2], b3 = (__bf16)v[3];
  const unsigned u0 = __builtin_bit_cast(unsigned short, b0), u1 = __builtin_bit_cast(unsigned short, b1);
  const unsigned u2 = __builtin_bit_cast(unsigned short, b2), u3 = __builtin_bit_cast(unsigned short, b3);
  return make_uint2(u0 | (u1 << 16), u2 | (u3 << 16));
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool NCHW_OUT, bool FASTK>
__global__ __launch_bounds__(kNT) void conv_gemm_bf16_kernel(ConvArgs p) {
  static_assert(WAVES_M * WAVES_N * 64 == kNT, "4 waves");
  constexpr int TM = BM / WAVES_M / 32;
  constexpr int TN = BN / WAVES_N / 32;
  constexpr int A_PER_T = BM / 16;      // 16 threads x 16 B (fp32) cover the 64 k of a row
  constexpr int B_PER_T = BN / 32;      //  8 threads x 16 B (bf16) cover the 64 k of a row
  constexpr int STAGE = (BM + BN) * kLdh;             // bf16 elements per LDS stage
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];

  const int L = xcd_swizzle(blockIdx.x, gridDim.x);
  const int mt = L / p.n_tiles, nt = L - mt * p.n_tiles;
  const int m0 = p.m_base + mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const int kqa = tid & 15, ra0 = tid >> 4;           // A: float4 group within the chunk / first row
  const int kqb = tid & 7, rb0 = tid >> 3;            // B: 8-bf16 group within the chunk / first row

  int a_y[A_PER_T], a_x[A_PER_T];
  unsigned a_pix[A_PER_T];
#pragma unroll
  for (int j = 0; j < A_PER_T; ++j) {
    const int pix = m0 + ra0 + 16 * j;
    const int b = pix / p.HW;
    const int rem = pix - b * p.HW;
    const int yy = rem / p.W;
    a_y[j] = pix < p.M ? yy : -(1 << 20);
    a_x[j] = rem - yy * p.W;
    a_pix[j] = (unsigned)pix;
  }
  const unsigned short* wrow = static_cast<const unsigned short*>(p.w) + (long long)(n0 + rb0) * p.Kpad + kqb * 8;
  const unsigned x_ld = (unsigned)p.x_ld;

  f32x4 ra[A_PER_T];
  uint4 rb[B_PER_T];
  unsigned a_mask = 0;
  unsigned a_off[A_PER_T];
  unsigned tap_mask = 0;
  int f_tap = 0, f_cc = 0;
  auto retap = [&]() {                  // FASTK (Cin % 64 == 0): a chunk is 64 channels of ONE tap
    const int ky = f_tap / p.ks, kx = f_tap - ky * p.ks;
    const int dy = ky - p.pad, dx = kx - p.pad;
    const int shift = dy * p.W + dx;
    tap_mask = 0;
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      const bool ok = (unsigned)(a_y[j] + dy) < (unsigned)p.H && (unsigned)(a_x[j] + dx) < (unsigned)p.W;
      a_off[j] = ok ? (a_pix[j] + (unsigned)shift) * x_ld + (unsigned)(kqa * 4) : (unsigned)(kqa * 4);
      tap_mask |= ok ? (1u << j) : 0u;
    }
  };
  if constexpr (FASTK) retap();

  auto gload = [&](int kc) {
    if constexpr (FASTK) {
      const unsigned coff = (unsigned)f_cc * kBKh;
#pragma unroll
      for (int j = 0; j < A_PER_T; ++j) ra[j] = *reinterpret_cast<const f32x4*>(p.x + (a_off[j] + coff));
      a_mask = tap_mask;
      if (++f_cc == p.cpt) {
        f_cc = 0;
        ++f_tap;
        retap();
      }
    } else {
      const int kg = kc * 16 + kqa;
      const int tap = kg / p.cpg;
      const int c4 = kg - tap * p.cpg;
      const int ky = tap / p.ks, kx = tap - ky * p.ks;
      const int dy = ky - p.pad, dx = kx - p.pad;
      const bool kvalid = kg < p.kgroups;
      a_mask = 0;
#pragma unroll
      for (int j = 0; j < A_PER_T; ++j) {
        const bool ok = kvalid && (unsigned)(a_y[j] + dy) < (unsigned)p.H && (unsigned)(a_x[j] + dx) < (unsigned)p.W;
        const unsigned off = ok ? (a_pix[j] + (unsigned)(dy * p.W + dx)) * x_ld + (unsigned)(c4 * 4) : 0u;
        ra[j] = *reinterpret_cast<const f32x4*>(p.x + off);
        a_mask |= ok ? (1u << j) : 0u;
      }
    }
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j)
      rb[j] = *reinterpret_cast<const uint4*>(wrow + (long long)j * 32 * p.Kpad + kc * kBKh);
  };
  auto sstore = [&](unsigned short* st) {        // fp32 -> bf16 (RNE) on the way into LDS
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      const uint2 v = (a_mask >> j) & 1u ? pack4(ra[j]) : make_uint2(0u, 0u);
      *reinterpret_cast<uint2*>(st + (ra0 + 16 * j) * kLdh + kqa * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < B_PER_T; ++j)
      *reinterpret_cast<uint4*>(st + (BM + rb0 + 32 * j) * kLdh + kqb * 8) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_off = (lane & 31) * kLdh + (lane >> 5) * 8;     // lane = row, 8 consecutive k per half-wave
  auto compute = [&](const unsigned short* st) {
    const unsigned short* sa = st + (wm * TM * 32) * kLdh + frag_off;
    const unsigned short* sb = st + (BM + wn * TN * 32) * kLdh + frag_off;
    bf16x8 af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(sa + i * 32 * kLdh);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const bf16x8*>(sb + j * 32 * kLdh);
#pragma unroll
    for (int s = 0; s < kBKh / 16; ++s) {
      if (s + 1 < kBKh / 16) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(sa + i * 32 * kLdh + (s + 1) * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(sb + j * 32 * kLdh + (s + 1) * 16);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = NCHW_OUT ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[s & 1][j], af[s & 1][i], acc[i][j], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0);
    }
  };

  gload(0);
  sstore(smem);
  __syncthreads();
  int cur = 0;
  for (int kc = 0; kc < p.nk; ++kc) {
    const bool more = kc + 1 < p.nk;
    if (more) gload(kc + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(smem + cur * STAGE);
    if (more) sstore(smem + (cur ^ 1) * STAGE);
    __syncthreads();
    cur ^= 1;
  }
  conv_epilogue<BM, BN, WAVES_M, WAVES_N, TM, TN, NCHW_OUT>(p, acc, reinterpret_cast<float*>(smem), m0, n0, mt, tid,
                                                             lane, wm, wn);
}

__global__ void pack_weight_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout, int cin,
                                        int ks, int mode, int rows_pad, int red4, int kpad) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)rows_pad * kpad) return;
  const int row = (int)(idx / kpad), k = (int)(idx - (long long)row * kpad);
  const int taps = ks * ks;
  const int tap = k / red4, r = k - tap * red4;
  const int rows = mode == 0 ? cout : cin, red = mode == 0 ? cin : cout;
  float v = 0.f;
  if (row < rows && tap < taps && r < red) {
    const int ky = tap / ks, kx = tap - ky * ks;
    if (mode == 0)
      v = w[(((long long)row * cin + r) * ks + ky) * ks + kx];
    else
      v = w[(((long long)r * cin + row) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
  }
  const __bf16 b = (__bf16)v;
  out[idx] = __builtin_bit_cast(unsigned short, b);
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <int BM, int BN, int WM, int WN>
int launch(const ConvArgs& a, bool nchw, hipStream_t stream) {
  const size_t lds = 2 * (size_t)(BM + BN) * kLdh * sizeof(unsigned short);
  const bool fast = a.cpt > 0;
  const dim3 grid(a.m_tiles * a.n_tiles), block(kNT);
#define FSD_LAUNCH(NCHW, FAST)                                                                                  \
  do {                                                                                                          \
    auto k = conv_gemm_bf16_kernel<BM, BN, WM, WN, NCHW, FAST>;                                                 \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return (int)e;                                                                         \
    fsd_prof::Scope prof(fsd_prof::kGemmBf16, 2.0 * ((double)a.M - a.m_base) * a.Cout * ((double)a.nk * kBKh), stream); \
    hipLaunchKernelGGL(k, grid, block, lds, stream, a);                                                         \
    return (int)hipGetLastError();                                                                              \
  } while (0)
  if (nchw) {
    if (fast) FSD_LAUNCH(true, true);
    FSD_LAUNCH(true, false);
  }
  if (fast) FSD_LAUNCH(false, true);
  FSD_LAUNCH(false, false);
#undef FSD_LAUNCH
}

constexpr int kBM = 128;
// narrow layers get narrow tiles so no MFMA columns (and no staged weight rows) are wasted
inline int tile_bn(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : 128); }

}  // namespace

extern "C" size_t fsd_packed_weight_elems_bf16(int rows, int red, int ksize) {
  return (size_t)round_up(rows, 128) * (size_t)round_up(ksize * ksize * round_up(red, 4), kBKh);
}

extern "C" int fsd_pack_conv_weight_bf16(const float* w_oihw, void* w_packed_bf16, int cout, int cin, int ksize,
                                         int mode, hipStream_t stream) {
  (void)hipGetLastError();
  if (!w_oihw || !w_packed_bf16 || cout < 1 || cin < 1 || (ksize != 1 && ksize != 3) || (mode != 0 && mode != 1))
    return FSD_ERR_ARG;
  const int rows = mode == 0 ? cout : cin, red = mode == 0 ? cin : cout;
  const int rows_pad = round_up(rows, 128), red4 = round_up(red, 4);
  const int kpad = round_up(ksize * ksize * red4, kBKh);
  const long long total = (long long)rows_pad * kpad;
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w_oihw,
                     static_cast<unsigned short*>(w_packed_bf16), cout, cin, ksize, mode, rows_pad, red4, kpad);
  return (int)hipGetLastError();
}

extern "C" int fsd_conv_row_tiles_bf16(long long pixels) { return (int)((pixels + kBM - 1) / kBM); }

extern "C" int fsd_conv2d_fwd_bf16(const float* x, long long x_ld, const void* w_packed_bf16, const float* bias,
                                   float* y, long long y_ld, float* bn_partial, int batch, int height, int width,
                                   int cin, int cout, int ksize, int out_nchw, hipStream_t stream) {
  (void)hipGetLastError();
  if (!x || !w_packed_bf16 || !y || batch < 1 || height < 1 || width < 1 || cout < 1) return FSD_ERR_ARG;
  if (ksize != 1 && ksize != 3) return FSD_ERR_UNSUPPORTED;
  if (cin < 4 || (cin & 3) || (x_ld & 3) || x_ld < cin) return FSD_ERR_ARG;
  if (!out_nchw && y_ld < cout) return FSD_ERR_ARG;
  if (out_nchw && bn_partial) return FSD_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w_packed_bf16) & 15)) return FSD_ERR_ARG;
  const long long pixels = (long long)batch * height * width;
  if (pixels > 0x7fffffffLL - 512) return FSD_ERR_UNSUPPORTED;
  if ((pixels + 1) * x_ld >= 0xffffffffLL) return FSD_ERR_UNSUPPORTED;
  ConvArgs a;
  a.x = x; a.w = w_packed_bf16; a.bias = bias; a.y = y; a.bn_partial = bn_partial;
  a.x_ld = x_ld; a.y_ld = y_ld;
  a.H = height; a.W = width; a.HW = height * width; a.M = (int)pixels;
  a.Cout = cout; a.ks = ksize; a.pad = (ksize - 1) / 2;
  a.cpg = cin / 4;
  a.kgroups = ksize * ksize * a.cpg;
  a.Kpad = round_up(ksize * ksize * cin, kBKh);
  a.nk = a.Kpad / kBKh;
  a.cpt = (cin % kBKh == 0) ? cin / kBKh : 0;
  const int bn = tile_bn(cout);
  a.m_tiles = (int)((pixels + kBM - 1) / kBM);
  a.n_tiles = (cout + bn - 1) / bn;
  a.m_base = 0;
  a.part_base = 0;
  a.batches = 1;
  a.x_bs = a.w_bs = a.y_bs = 0;
  if (bn == 32) return launch<kBM, 32, 4, 1>(a, out_nchw != 0, stream);
  if (bn == 64) return launch<kBM, 64, 2, 2>(a, out_nchw != 0, stream);
  return launch<kBM, 128, 2, 2>(a, out_nchw != 0, stream);
}
