// Direct 3x3 convolution of the NARROW layers (32 / 64 channels in, 32 / 64 out: darknet L2 at 208x208, the reweighting net's
// second layer, and their data gradients) under the split arithmetic (conv.hip: fp32 operands as three bf16 planes, six
// v_mfma_f32_32x32x16_bf16 terms per product, fp32 accumulate).
//
// conv_gemm_kernel walks K = taps x Cin chunk by chunk and stages, per 32-channel chunk, the 128 input rows of ONE tap: every
// input pixel is fetched and SPLIT nine times per workgroup (once per tap), and so is every weight -- with 64 output channels
// there are only 24 MFMAs per wave to hide ~250 staging instructions behind (measured: 32->64 at 208x208, 0.98 ms = 0.62 PF of
// bf16 issue; PMC MFMA-busy 0.34-0.37 on these launches).  Here a workgroup owns an 8 x 16 block of output pixels:
//   * the 10 x 18 HALO patch of its input (one 32-channel slice) is fetched and split ONCE into three bf16 planes in LDS
//     (unpadded 64-byte pixel rows, 16-byte pieces XOR-permuted by (row >> 2) & 3); the A fragment of tap (dy, dx) is the
//     same ds_read_b128 at a pixel offset of dy * 18 + dx.  The MFMA row m of a wave's 2 x 16 pixels is mapped to (row, column)
//     so that each 16-lane service group of the read covers 16 CONSECUTIVE halo pixels (conflict-free for every tap);
//   * the weights of one KERNEL ROW (3 taps x Cout x 32 channels) sit in LDS as three planes; the next row's are fetched
//     during the current row's MFMAs and split one micro-step behind each MFMA (conv_gemm_split8_kernel's schedule);
//   * per wave and kernel row: 36 x Cout/32 MFMAs from 9 + 9 x Cout/32 fragment reads.
// 4 waves, ~70 KB of LDS: two workgroups per CU hide each other's patch load and epilogue.
// Epilogue: the 128 x Cout tile crosses LDS once and leaves as 16-byte stores (NHWC), + bias, optional leaky, and the
// per-block BatchNorm partial sums (one partial row per workgroup = per 128 pixels, the row count fsd_conv_row_tiles
// reports for these shapes).  Shapes: H % 8 == 0, W % 16 == 0, Cin in {32, 64}, Cout in {32, 64}; everything else stays on
// conv_gemm_kernel.  FSD_CONV_HALO=0 switches it off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fsdet.h"
#include "conv_common.hpp"
#include "profile.hpp"

namespace {

using namespace fsd_conv;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kBH = 8, kBW = 16;                   // output block
constexpr int kHW = kBW + 2, kHH = kBH + 2;        // halo patch 18 x 10
constexpr int kHaloPx = kHW * kHH;                 // 180 pixels
constexpr int kRowB = 64;                          // bytes of one plane row: 32 bf16
constexpr int kPlaneA = kHaloPx * kRowB;           // 11520 B

struct HaloArgs {
  const float* x;
  const float* w;        // packed fp32 weights [Cout up to 128][9 * Cin] (fsd_pack_conv_weight, either mode)
  const float* bias;
  float* y;
  float* bn_partial;     // [blocks][Cout][2] or null
  long long x_ld, y_ld;
  int B, H, W, Cin, Cout, Kpad;
  int bx, by;            // blocks per image row / column
  float slope;
};

__device__ __forceinline__ unsigned cvt2(float a, float b) {      // one v_cvt_pk_bf16_f32
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_f(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

// MFMA row m (0..31) of a wave -> (row, column) of its 2 x 16 output pixels.  The ds_read_b128 of a wave is served in
// 16-lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): the first group takes pixel row 0, the second row 1, so a
// group reads 16 consecutive halo pixels.
__device__ __forceinline__ void pix_of(int m, int& py, int& px) {
  const bool g1 = (m >= 4 && m < 12) || (m >= 16 && m < 20) || m >= 28;
  py = g1 ? 1 : 0;
  px = m < 4 ? m : m < 12 ? m - 4 : m < 20 ? m - 8 : m < 28 ? m - 12 : m - 16;
}

template <int COUT>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(HaloArgs p) {
  constexpr int TN = COUT / 32;
  constexpr int B_F4 = 3 * COUT * 8 / 256;          // float4 of one kernel row of weights per thread (64: 6, 32: 3)
  constexpr int PLANE_B = COUT * kRowB;             // one plane of one tap
  constexpr int MF_ROW = 3 * 2 * 6 * TN;            // MFMAs per wave and kernel row
  static_assert(B_F4 * 8 <= MF_ROW, "the split micro-steps of the next kernel row fit behind this row's MFMAs");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  unsigned char* sA = smem_h;                       // [3 planes][180 px][64 B]
  unsigned char* sB = smem_h + 3 * kPlaneA;         // [3 taps][3 planes][COUT][64 B]

  const int blk = xcd_swizzle(blockIdx.x, gridDim.x);
  const int bxi = blk % p.bx;
  const int t2 = blk / p.bx;
  const int byi = t2 % p.by;
  const int img = t2 / p.by;
  const int x0 = bxi * kBW, y0 = byi * kBH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = tid & 7;
  const int slices = p.Cin / 32;

  // ---- staging roles ----
  // A: halo pixel e >> 3, channel group e & 7 for e = tid + 256 i
  auto a_src = [&](int i, int slice, bool& ok) -> const float* {
    const int e = tid + 256 * i, hp = e >> 3;
    const int hy = hp / kHW, hx = hp - hy * kHW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    ok = hp < kHaloPx && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    return p.x + ((long long)(img * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)) * p.x_ld + slice * 32 + kq * 4;
  };
  auto a_dst = [&](int i) -> int {                  // byte offset inside a plane
    const int hp = (tid + 256 * i) >> 3;
    return hp * kRowB + (((kq >> 1) ^ ((hp >> 2) & 3)) << 4) + (kq & 1) * 8;
  };
  // B: (tap of the row, output channel n) = e >> 3 for e = tid + 256 i
  auto b_src = [&](int i, int krow, int slice) -> const float* {
    const int e = tid + 256 * i, tn = e >> 3;
    const int tl = tn / COUT, n = tn - tl * COUT;
    return p.w + (long long)n * p.Kpad + (krow * 3 + tl) * p.Cin + slice * 32 + kq * 4;
  };
  auto b_dst = [&](int i) -> int {                  // byte offset of the first plane's piece inside sB
    const int tn = (tid + 256 * i) >> 3;
    const int tl = tn / COUT, n = tn - tl * COUT;
    return tl * 3 * PLANE_B + n * kRowB + (((kq >> 1) ^ ((n >> 2) & 3)) << 4) + (kq & 1) * 8;
  };

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- fragment geometry ----
  int fpy, fpx;
  pix_of(lane & 31, fpy, fpx);
  const int hr0 = (2 * wave + fpy) * kHW + fpx;     // halo pixel of tap (0, 0) for this lane's output pixel
  const int fh = lane >> 5;
  const int nrow = lane & 31, nsw = (nrow >> 2) & 3;

  f32x4 rb[B_F4];
  auto load_b = [&](int krow, int slice) {
#pragma unroll
    for (int i = 0; i < B_F4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(b_src(i, krow, slice));
  };
  auto store_b_now = [&]() {                        // un-hidden form (prologue of a slice)
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      uint2 h, m, l;
      split3(rb[i], h, m, l);
      unsigned char* d = sB + b_dst(i);
      *reinterpret_cast<uint2*>(d) = h;
      *reinterpret_cast<uint2*>(d + PLANE_B) = m;
      *reinterpret_cast<uint2*>(d + 2 * PLANE_B) = l;
    }
  };
  auto stage_a = [&](int slice) {
    f32x4 ra[6];
    bool ok[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float* s = a_src(i, slice, ok[i]);
      ra[i] = ok[i] ? *reinterpret_cast<const f32x4*>(s) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (((tid + 256 * i) >> 3) < kHaloPx) {
        uint2 h, m, l;
        split3(ra[i], h, m, l);
        unsigned char* d = sA + a_dst(i);
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + kPlaneA) = m;
        *reinterpret_cast<uint2*>(d + 2 * kPlaneA) = l;
      }
    }
  };

  constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // the six terms, smallest first
  for (int slice = 0; slice < slices; ++slice) {
    if (slice > 0) __syncthreads();                 // every wave is done reading the previous slice
    load_b(0, slice);
    stage_a(slice);
    store_b_now();
    __syncthreads();
    for (int krow = 0; krow < 3; ++krow) {
      const bool more = krow < 2;
      if (more) load_b(krow + 1, slice);
      unsigned sh[B_F4][2], sm_[B_F4][2], sl[B_F4][2];      // planes of the next kernel row's weights, formed below
      float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
      bf16x8 af[3], bf[3][TN];
      auto frags = [&](int tl, int ks) {
        const int hr = hr0 + krow * kHW + tl;
        const int po = hr * kRowB + ((((2 * ks + fh)) ^ ((hr >> 2) & 3)) << 4);
        const int pb = tl * 3 * PLANE_B + nrow * kRowB + (((2 * ks + fh) ^ nsw) << 4);
        constexpr int QA[3] = {0, 2, 1}, QB[3] = {2, 0, 1};       // first-use order of the six terms
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          af[QA[o]] = *reinterpret_cast<const bf16x8*>(sA + QA[o] * kPlaneA + po);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            bf[QB[o]][j] = *reinterpret_cast<const bf16x8*>(sB + QB[o] * PLANE_B + pb + j * 32 * kRowB);
        }
      };
#pragma unroll
      for (int g = 0; g < 6; ++g) {                 // (tap of the row, k-step): 6 terms x TN MFMAs each
        const int tl = g >> 1, ks = g & 1;
        frags(tl, ks);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA[t]], bf[TB[t]][j], acc[j], 0, 0, 0);
            // split of the next row's weights: micro-step u - (MF_ROW - 8 B_F4) behind the LAST 8 B_F4 MFMAs of the row
            const int u = (g * 6 + t) * TN + j, ms = u - (MF_ROW - 8 * B_F4);
            if (more && ms >= 0) {
              const int f = ms / 8, step = ms % 8;
              const f32x4 v = rb[f];
              if (step == 0) { sh[f][0] = cvt2(v[0], v[1]); sh[f][1] = cvt2(v[2], v[3]); }
              else if (step == 1) { r0 = v[0] - lo_f(sh[f][0]); r1 = v[1] - hi_f(sh[f][0]); }
              else if (step == 2) { r2 = v[2] - lo_f(sh[f][1]); r3 = v[3] - hi_f(sh[f][1]); }
              else if (step == 3) { sm_[f][0] = cvt2(r0, r1); sm_[f][1] = cvt2(r2, r3); }
              else if (step == 4) { r0 -= lo_f(sm_[f][0]); r1 -= hi_f(sm_[f][0]); }
              else if (step == 5) { r2 -= lo_f(sm_[f][1]); r3 -= hi_f(sm_[f][1]); }
              else if (step == 6) { sl[f][0] = cvt2(r0, r1); sl[f][1] = cvt2(r2, r3); }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
      }
      if (more) {
        __syncthreads();                            // every wave has read this row's weights
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
          unsigned char* d = sB + b_dst(i);
          *reinterpret_cast<uint2*>(d) = make_uint2(sh[i][0], sh[i][1]);
          *reinterpret_cast<uint2*>(d + PLANE_B) = make_uint2(sm_[i][0], sm_[i][1]);
          *reinterpret_cast<uint2*>(d + 2 * PLANE_B) = make_uint2(sl[i][0], sl[i][1]);
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();                                  // staging LDS is free: the epilogue re-uses it

  // ---- epilogue ----
  float* tile = reinterpret_cast<float*>(smem_h);   // [128 px][COUT]
  const int c_lane = lane & 31, r_lane = 4 * (lane >> 5);
  if (p.bn_partial != nullptr) {
    float* s_stat = tile + 128 * COUT;              // [4 waves][COUT][2] behind the tile
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[j][r];
        s += v;
        q += v * v;
      }
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      if (lane < 32) {
        s_stat[(wave * COUT + j * 32 + c_lane) * 2 + 0] = s;
        s_stat[(wave * COUT + j * 32 + c_lane) * 2 + 1] = q;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = j * 32 + c_lane;
    const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int py, px;
      pix_of((r & 3) + 8 * (r >> 2) + r_lane, py, px);
      float v = acc[j][r] + bv;
      if (p.slope != 1.f) v = v > 0.f ? v : v * p.slope;
      tile[((2 * wave + py) * kBW + px) * COUT + n] = v;
    }
  }
  __syncthreads();
  if (p.bn_partial != nullptr && tid < COUT) {
    const float* s_stat = tile + 128 * COUT;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      s += s_stat[(w * COUT + tid) * 2 + 0];
      q += s_stat[(w * COUT + tid) * 2 + 1];
    }
    float* dst = p.bn_partial + ((long long)blk * p.Cout + tid) * 2;
    dst[0] = s;
    dst[1] = q;
  }
  constexpr int PPR = COUT / 4;                     // float4 pieces per pixel
#pragma unroll
  for (int it = tid; it < 128 * PPR; it += 256) {
    const int pxl = it / PPR, pc = it - pxl * PPR;
    const int oy = y0 + pxl / kBW, ox = x0 + pxl % kBW;
    *reinterpret_cast<f32x4*>(p.y + ((long long)(img * p.H + oy) * p.W + ox) * p.y_ld + pc * 4) =
        *reinterpret_cast<const f32x4*>(tile + pxl * COUT + pc * 4);
  }
}

}  // namespace

bool fsd_conv::halo_ok(int height, int width, int cin, int cout, int ksize, bool nchw) {
  static const char* env = getenv("FSD_CONV_HALO");
  if (env && env[0] == '0') return false;
  return fsd_conv::f32_split_on() && ksize == 3 && !nchw && (cin == 32 || cin == 64) && (cout == 32 || cout == 64) &&
         height % kBH == 0 && width % kBW == 0;
}

int fsd_conv::conv3x3_halo(const float* x, long long x_ld, const float* w_packed, int kpad, const float* bias, float* y,
                           long long y_ld, float* bn_partial, int batch, int height, int width, int cin, int cout, float slope,
                           hipStream_t stream) {
  if ((y_ld & 3) || (reinterpret_cast<uintptr_t>(y) & 15)) return FSD_ERR_UNSUPPORTED;
  HaloArgs a;
  a.x = x; a.w = w_packed; a.bias = bias; a.y = y; a.bn_partial = bn_partial;
  a.x_ld = x_ld; a.y_ld = y_ld;
  a.B = batch; a.H = height; a.W = width; a.Cin = cin; a.Cout = cout; a.Kpad = kpad;
  a.bx = width / kBW; a.by = height / kBH;
  a.slope = slope;
  const long long blocks = (long long)batch * a.bx * a.by;
  if (blocks > 0x7fffffffLL) return FSD_ERR_UNSUPPORTED;
  const size_t lds_stage = 3 * (size_t)kPlaneA + 9 * (size_t)cout * kRowB;
  const size_t lds_tile = (size_t)128 * cout * 4 + 4 * (size_t)cout * 2 * 4;
  const size_t lds = lds_stage > lds_tile ? lds_stage : lds_tile;
  // issued MFMA work: every output pixel x column x (tap, channel)
  fsd_prof::Scope prof(fsd_prof::kGemmFwd, 2.0 * (double)batch * height * width * cout * 9.0 * cin, stream);
  if (cout == 64) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<64>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    FSD_LAUNCH(conv3x3_halo_kernel<64>, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  } else {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    FSD_LAUNCH(conv3x3_halo_kernel<32>, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  }
  return (int)hipGetLastError();
}
