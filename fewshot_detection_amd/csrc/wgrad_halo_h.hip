// bf16 storage mode: weight gradient of the NARROW 3x3 layers (32 -> 64: darknet L2 at 208x208; 64 -> 128: L4 / L6 at
// 104x104; their twins in the reweighting net), dy and x bf16 NHWC, fp32 accumulate on v_mfma_f32_32x32x16_bf16.
//
//   dW[co][tap][ci] = sum over pixels p of  dy[p][co] * x[p + tap][ci]          M = Cout, N = 9 x Cin, K = B*H*W pixels
//
// wgrad_bf16_tr_kernel treats this as a GEMM whose N tiles are taps: every workgroup re-stages dy for each tap group and a
// shifted copy of x per tap, 32 pixels per barrier -- 0.29 ms (32 -> 64) and 0.24 ms (64 -> 128) at B = 64 for layers whose
// operands are 0.11 / 0.05 ms of HBM time.  Here (wgrad_halo.hip's idea for the fp32 mode, with LDS-DMA staging):
//   * a PERSISTENT workgroup walks a contiguous run of 8 x 16 pixel blocks with its accumulators in registers;
//   * per block the dy tile (128 px x Cout) and the 10 x 18 halo patch of x are staged ONCE, as they lie ([pixel][channel]
//     rows), by LDS-DMA, double-buffered; all nine taps read the same patch: the B fragment of tap (ty, tx) for the block row
//     s is the transposing read ds_read_b64_tr_b16 at patch pixel (s + ty) * 18 + tx (TileGeom's row swizzle is keyed on
//     row & 3, so any pixel offset is conflict-free);
//   * 32 -> 64: six waves, wave = (channel tile of dy, kernel row): 3 accumulators; 64 -> 128: eight waves, wave = (channel
//     tile of dy, 32-channel half of x): 9 accumulators.  A k-step = one block row (16 pixels): 1 + NACC fragments feed
//     NACC MFMAs;
//   * one [Cout][9 Cin] partial per workgroup, folded by wgrad_h_fold_kernel in a fixed order (conv_bf16v2.hip).
// Shapes: ksize 3, (Cin, Cout) = (32, 64) or (64, 128), H % 8 == 0, W % 8 == 0 (104 x 104: the last block of a row is half
// outside, its dy pixels are staged as zeros).  FSD_CONV_HALO=0 switches it off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "fsdet.h"
#include "conv_common.hpp"
#include "profile.hpp"

namespace {

using namespace fsd_conv;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int kBH = 8, kBW = 16;
constexpr int kHW = kBW + 2, kHH = kBH + 2;
constexpr int kHaloPx = kHW * kHH;                 // 180
constexpr int kBlkPx = kBH * kBW;                  // 128

__device__ __attribute__((aligned(16))) u16 g_zero_page_wh[64];

struct WgradHaloHArgs {
  const u16* dy; const u16* x;
  float* ws;                 // [gridDim.x][Cout][9 * Cin]
  unsigned dy_ld, x_ld;
  int H, W, bx, by, blocks;
};

__device__ __forceinline__ void dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// XOR mask on the 16-byte piece index of a [pixel][CH channels] row (conv_bf16v2.hip TileGeom: conflict-free transpose reads)
template <int CH>
__device__ __forceinline__ int swz_of(int row) { return CH == 128 ? ((row & 3) << 2) : CH == 64 ? ((row & 2) << 1) : 0; }

template <int CIN, int COUT>
__global__ __launch_bounds__(CIN == 32 ? 384 : 512, 1) void wgrad3x3_halo_h_kernel(WgradHaloHArgs p) {
  static_assert((CIN == 32 && COUT == 64) || (CIN == 64 && COUT == 128), "32 -> 64 and 64 -> 128");
  constexpr int NT = CIN == 32 ? 384 : 512;
  constexpr int NACC = CIN == 32 ? 3 : 9;           // accumulators of a wave
  constexpr int DPR = COUT / 8, XPR = CIN / 8;      // 16-byte pieces per dy / x pixel row
  constexpr int D_PIECES = kBlkPx * DPR, X_PIECES = kHaloPx * XPR;
  constexpr int D_IT = (D_PIECES + NT - 1) / NT, X_IT = (X_PIECES + NT - 1) / NT;
  constexpr int D_BUF = D_IT * NT * 16, X_BUF = X_IT * NT * 16;      // bytes (tails are zeros)
  constexpr int STAGE = D_BUF + X_BUF;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_wh[];      // [2][STAGE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 32 -> 64: mt = wave & 1, kernel row = wave >> 1;  64 -> 128: mt = wave & 3, x half = wave >> 2
  const int mt = CIN == 32 ? (wave & 1) : (wave & 3);
  const int sel = CIN == 32 ? (wave >> 1) : (wave >> 2);
  const int G = lane >> 4, Lq = lane & 15;

  const int nwg = gridDim.x, g = blockIdx.x;
  const int b_begin = (int)((long long)p.blocks * g / nwg), b_end = (int)((long long)p.blocks * (g + 1) / nwg);

  // ---- staging roles (block-independent): piece e = tid + NT i ----
  int d_rel[D_IT], d_col[D_IT], x_rel[X_IT];
  unsigned x_hyx[X_IT];
#pragma unroll
  for (int i = 0; i < D_IT; ++i) {
    const int e = tid + NT * i, px = e / DPR, q = e - px * DPR;
    const int py = px >> 4, pxx = px & 15;
    const int piece = q ^ swz_of<COUT>(px);
    d_rel[i] = (py * p.W + pxx) * (int)p.dy_ld + piece * 8;
    d_col[i] = px < kBlkPx ? pxx : 1 << 20;         // pixel column inside the block (tail pieces: never inside the image)
  }
#pragma unroll
  for (int i = 0; i < X_IT; ++i) {
    const int e = tid + NT * i, hp = e / XPR, q = e - hp * XPR;
    const int hy = hp / kHW, hx = hp - hy * kHW;
    const int piece = q ^ swz_of<CIN>(hp);
    x_rel[i] = ((hy - 1) * p.W + (hx - 1)) * (int)p.x_ld + piece * 8;
    x_hyx[i] = hp < kHaloPx ? (unsigned)((hy << 8) | hx) : 0xffffu;
  }
  const u16* zero_src = g_zero_page_wh;
  asm volatile("" : "+v"(zero_src));

  int bxi, byi, img;                                // block being FETCHED (carried, no divisions in the loop)
  {
    bxi = b_begin % p.bx;
    const int t2 = b_begin / p.bx;
    byi = t2 % p.by;
    img = t2 / p.by;
  }
  auto stage = [&](bool live, unsigned char* st) {
    const int y0 = byi * kBH, x0 = bxi * kBW;
    const long long pix0 = (long long)(img * p.H + y0) * p.W + x0;
    const u16* dbase = p.dy + pix0 * p.dy_ld;
    const u16* xbase = p.x + pix0 * p.x_ld;
#pragma unroll
    for (int i = 0; i < D_IT; ++i) {
      int rel = d_rel[i];
      asm volatile("" : "+v"(rel));
      // (W % 16 == 8: the last block of a row is half outside -- those dy pixels are staged as zeros and drop out of the sums)
      dma16(live && x0 + d_col[i] < p.W ? dbase + rel : zero_src, st + (NT * i + wave * 64) * 16);
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      int rel = x_rel[i];
      asm volatile("" : "+v"(rel));
      const int hy = (int)(x_hyx[i] >> 8), hx = (int)(x_hyx[i] & 255u);
      const bool ok = live && x_hyx[i] != 0xffffu && (unsigned)(y0 - 1 + hy) < (unsigned)p.H &&
                      (unsigned)(x0 - 1 + hx) < (unsigned)p.W;
      dma16(ok ? xbase + rel : zero_src, st + D_BUF + (NT * i + wave * 64) * 16);
    }
    if (++bxi == p.bx) {
      bxi = 0;
      if (++byi == p.by) { byi = 0; ++img; }
    }
  };

  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  // transposing fragment read of a 32-channel block at pixel row `krow` of a [pixel][CH] tile: 16-lane group G reads rows
  // krow + 8 (G >> 1) + (Lq >> 2) (+4), channels ch0 + 16 (G & 1) + 4 (Lq & 3)   (wgrad_bf16_tr_kernel's frag)
  auto frag = [&](const unsigned char* tile, auto chc, int ch0, int krow) -> bf16x8 {
    constexpr int CH = decltype(chc)::value;
    const int row = krow + (G >> 1) * 8 + (Lq >> 2);
    const int ch = ch0 + 16 * (G & 1) + 4 * (Lq & 3);
    const int pc = (ch >> 3) ^ swz_of<CH>(row);
    const u16* a = reinterpret_cast<const u16*>(tile) + row * CH + pc * 8 + (ch & 7);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a + 4 * CH));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  if (b_begin < b_end) {
    stage(true, s_wh);
    int cur = 0;
    for (int blk = b_begin; blk < b_end; ++blk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                              // this block's operands have landed; the other stage is free
      if (blk + 1 < b_end) stage(true, s_wh + (cur ^ 1) * STAGE);
      const unsigned char* sD = s_wh + cur * STAGE;
      const unsigned char* sX = sD + D_BUF;
#pragma unroll
      for (int s = 0; s < kBH; ++s) {               // k-step = block row s (16 pixels)
        const bf16x8 af = frag(sD, std::integral_constant<int, COUT>(), mt * 32, s * kBW);
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
          const int tap = CIN == 32 ? sel * 3 + a : a;
          const int ty = tap / 3, tx = tap - 3 * ty;
          const bf16x8 bf = frag(sX, std::integral_constant<int, CIN>(), CIN == 32 ? 0 : sel * 32, (s + ty) * kHW + tx);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[a], 0, 0, 0);
        }
      }
      cur ^= 1;
    }
  }
  // ---- this workgroup's partial: ws[g][co][tap * CIN + ci] ----
  float* out = p.ws + (long long)g * COUT * 9 * CIN;
  const int c_lane = lane & 31, r_lane = 4 * (lane >> 5);
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    const int tap = CIN == 32 ? sel * 3 + a : a;
    const int n = tap * CIN + (CIN == 32 ? 0 : sel * 32) + c_lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + r_lane;
      out[m * 9 * CIN + n] = acc[a][r];
    }
  }
}

}  // namespace

bool fsd_conv::wgrad_halo_h_ok(int height, int width, int cin, int cout, int ksize) {
  static const char* env = getenv("FSD_CONV_HALO");
  if (env && env[0] == '0') return false;
  return ksize == 3 && ((cin == 32 && cout == 64) || (cin == 64 && cout == 128)) && height % kBH == 0 && width % 8 == 0;
}

// workgroups = partial slices of the fold: two 6-wave workgroups per CU (32 -> 64), one 8-wave workgroup per CU (64 -> 128)
int fsd_conv::wgrad_halo_h_slots(int batch, int height, int width, int cin) {
  const long long blocks = (long long)batch * (height / kBH) * ((width + kBW - 1) / kBW);
  const int wgs = cin == 32 ? 512 : 256;
  return (int)(blocks < wgs ? blocks : wgs);
}

int fsd_conv::wgrad3x3_halo_h(const void* dy, long long dy_ld, const void* x, long long x_ld, float* ws, int batch, int height,
                              int width, int cin, int cout, hipStream_t stream) {
  const long long pixels = (long long)batch * height * width;
  if ((dy_ld & 7) || (x_ld & 7) || (reinterpret_cast<uintptr_t>(dy) & 15) || (reinterpret_cast<uintptr_t>(x) & 15) ||
      pixels >= 0x7fffffffLL || (8LL * width + 16) * dy_ld >= 0x7fffffffLL || (10LL * width + 18) * x_ld >= 0x7fffffffLL)
    return FSD_ERR_UNSUPPORTED;
  WgradHaloHArgs a;
  a.dy = static_cast<const u16*>(dy); a.x = static_cast<const u16*>(x); a.ws = ws;
  a.dy_ld = (unsigned)dy_ld; a.x_ld = (unsigned)x_ld;
  a.H = height; a.W = width; a.bx = (width + kBW - 1) / kBW; a.by = height / kBH;
  a.blocks = batch * a.bx * a.by;
  const int wgs = wgrad_halo_h_slots(batch, height, width, cin);
  fsd_prof::Scope prof(fsd_prof::kGemmBf16, 2.0 * (double)pixels * cout * 9.0 * cin, stream);
  if (cin == 32) {
    // stage = 128 px x 128 B (dy: 1024 pieces = 3 passes of 384) + 180 px x 64 B (x: 720 pieces = 2 passes)
    const int lds = 2 * (3 * 384 * 16 + 2 * 384 * 16);
    auto k = wgrad3x3_halo_h_kernel<32, 64>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    FSD_LAUNCH(k, dim3(wgs), dim3(384), lds, stream, a);
  } else {
    // stage = 128 px x 256 B (dy: 2048 pieces = 4 passes of 512) + 180 px x 128 B (x: 1440 pieces = 3 passes)
    const int lds = 2 * (4 * 512 * 16 + 3 * 512 * 16);
    auto k = wgrad3x3_halo_h_kernel<64, 128>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    FSD_LAUNCH(k, dim3(wgs), dim3(512), lds, stream, a);
  }
  return (int)hipGetLastError();
}
