// Weight gradient of the NARROW 3x3 layers (32 input channels, 64 output channels: darknet L2 at 208x208 and the reweighting
// net's second layer) under the split arithmetic (conv.hip: fp32 operands as three bf16 planes, six v_mfma_f32_32x32x16_bf16
// terms per product, fp32 accumulate).
//
//   dW[co][tap][ci] = sum over pixels p of  dy[p][co] * x[p + tap][ci]          M = 64, N = 9 x 32, K = B*H*W pixels
//
// wgrad_kernel<64, ...> (wgrad.hip) treats this as a 64 x 288 GEMM in five 64 x 64 tiles: every workgroup re-reads and re-splits
// the whole dy (five times in all) and a tap pair's shifted copy of x, 11.4 issued instructions per MFMA -- 1.10 ms at B = 64
// (93 TFLOP/s of fp32 work, 0.22 of the six-term ceiling; the layer's operands are 1.06 GB = 0.2 ms of HBM time).  Here a
// workgroup owns 8 x 8 pixel blocks (conv_halo.hip's idea, transposed):
//   * the block's dy (64 px x 64 ch) and the 10 x 10 HALO patch of x (100 px x 32 ch) are fetched ONCE, split once into three
//     bf16 planes and stored in LDS as they lie ([pixel][channel] rows of 128 / 64 bytes); all nine taps read the same patch:
//     the B fragment of tap (ty, tx) is the transposing read ds_read_b64_tr_b16 at a pixel offset of ty * 10 + tx;
//   * six waves: wave w owns output channels 32 (w & 1) .. + 31 and kernel row w >> 1 (three taps = three 32 x 32
//     accumulators); a k-step = two block rows (16 pixels), per k-step 6 + 18 fragment reads feed 18 MFMAs;
//   * a workgroup walks a contiguous run of blocks with its accumulators in registers (the next block's operands are in flight
//     during the MFMAs) and stores ONE 64 x 288 partial; wgrad_reduce_kernel folds the partials in a fixed order.
// 44 KB of LDS, 128 registers: two workgroups per CU.  Measured at B = 64, 208x208 (tools/layer_bench.py wgrad): 0.67 ms against
// 1.00 for wgrad_kernel (in the step 0.83 -> see DESIGN.md).  What the timing experiments of round 5 say about the rest
// (tools/experiments_r05/gpu_r05g.sh, phases removed one at a time): the MFMAs alone run at their floor (0.35 ms), the transposing
// fragment reads + barriers take 0.19, the split + LDS stores 0.11, the global loads 0.24 -- and the four ADD UP: one workgroup
// per CU (256 instead of 512 workgroups) is only 1.2x slower, i.e. two co-resident workgroups hardly overlap their phases.
// Shapes: Cin = 32, Cout = 64, H % 8 == 0, W % 8 == 0; everything else stays on wgrad_kernel.
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

constexpr int kPB = 8;                       // pixel block 8 x 8
constexpr int kHP = kPB + 2;                 // halo 10 x 10
constexpr int kHaloPx = kHP * kHP;           // 100
constexpr int kBlkPx = kPB * kPB;            // 64
constexpr int kCin = 32, kCout = 64;
constexpr int kXPlane = kHaloPx * kCin;      // bf16 elements of one x plane
constexpr int kDPlane = kBlkPx * kCout;      // bf16 elements of one dy plane
constexpr int kThreadsH = 384;
constexpr int kXF4 = (kHaloPx * (kCin / 4) + kThreadsH - 1) / kThreadsH;     // float4 of x per thread and block: 3
constexpr int kDF4 = (kBlkPx * (kCout / 4) + kThreadsH - 1) / kThreadsH;     // float4 of dy per thread and block: 3
constexpr size_t kLdsH = (size_t)3 * (kXPlane + kDPlane) * sizeof(u16);       // 43776 B

struct WgradHaloArgs {
  const float* dy; const float* x; float* ws;        // ws: [blocks][64][288]
  long long dy_ld, x_ld;
  int B, H, W;
  int bx, by;                                        // 8 x 8 blocks per image row / column
  long long patches;                                 // B * by * bx
  int per;                                           // patches per workgroup
};

__global__ __launch_bounds__(kThreadsH, 4) void wgrad3x3_halo_kernel(WgradHaloArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_wh[];
  u16* sX = reinterpret_cast<u16*>(smem_wh);                 // [3 planes][100 px][32 ch]
  u16* sD = sX + 3 * kXPlane;                                // [3 planes][64 px][64 ch], 16-byte pieces XOR-permuted per row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mt = wave & 1, krow = wave >> 1;
  const long long pt_begin = (long long)blockIdx.x * p.per;
  long long pt_end = pt_begin + p.per;
  if (pt_end > p.patches) pt_end = p.patches;

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- staging: global -> registers (one block ahead) -> three bf16 planes in LDS ----
  // Addresses: everything that depends on the thread is computed ONCE -- the element offset of each of its six 16-byte pieces from
  // the block's first pixel and which image border the piece lies beyond if the block touches that border (bits: 1 top, 2 bottom,
  // 4 left, 8 right; 16 = no piece) -- and a block costs a uniform base pointer, four uniform edge flags and six adds.  (First
  // version: 64-bit divisions of the block index and full per-piece coordinate arithmetic for every block -- ~450 of the ~860
  // instructions a wave issued per block, for 72 MFMAs: the kernel ran at the same speed with its MFMAs removed.)
  // Loads are unconditional from clamped addresses; the zero-select of the border happens in sstore (a select right behind its
  // load makes the compiler wait for the data inside gload).
  int xrel[kXF4], drel[kDF4];
  unsigned xedge[kXF4];
#pragma unroll
  for (int i = 0; i < kXF4; ++i) {
    const int e = tid + kThreadsH * i, hp = e >> 3, kq = e & 7;
    const int hy = hp / kHP, hx = hp - hy * kHP;
    xrel[i] = ((hy - 1) * p.W + (hx - 1)) * (int)p.x_ld + kq * 4;
    xedge[i] = hp >= kHaloPx ? 16u : (hy == 0 ? 1u : 0u) | (hy == kHP - 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == kHP - 1 ? 8u : 0u);
  }
#pragma unroll
  for (int i = 0; i < kDF4; ++i) {
    const int e = tid + kThreadsH * i, px = e >> 4, cq = e & 15;
    drel[i] = px < kBlkPx ? ((px >> 3) * p.W + (px & 7)) * (int)p.dy_ld + cq * 4 : 0;
  }
  // block position of the next gload (uniform), advanced block by block
  int g_bx = (int)(pt_begin % p.bx);
  int g_by = (int)((pt_begin / p.bx) % p.by);
  int g_img = (int)(pt_begin / ((long long)p.bx * p.by));
  f32x4 rx[kXF4], rd[kDF4];
  unsigned xmask = 0;
  auto gload = [&]() {
    const long long org = ((long long)g_img * p.H + g_by * kPB) * p.W + g_bx * kPB;        // first pixel of the block
    const float* xb = p.x + org * p.x_ld;
    const float* db = p.dy + org * p.dy_ld;
    const unsigned at = (g_by == 0 ? 1u : 0u) | (g_by == p.by - 1 ? 2u : 0u) | (g_bx == 0 ? 4u : 0u) | (g_bx == p.bx - 1 ? 8u : 0u) | 16u;
    xmask = 0;
#pragma unroll
    for (int i = 0; i < kXF4; ++i) {
      const bool ok = (xedge[i] & at) == 0;
      rx[i] = *reinterpret_cast<const f32x4*>(xb + (ok ? xrel[i] : 0));
      xmask |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < kDF4; ++i) rd[i] = *reinterpret_cast<const f32x4*>(db + drel[i]);
    if (++g_bx == p.bx) {
      g_bx = 0;
      if (++g_by == p.by) { g_by = 0; ++g_img; }
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < kXF4; ++i) {
      const int e = tid + kThreadsH * i, hp = e >> 3, kq = e & 7;
      if (hp < kHaloPx) {
        uint2 h, m, l;
        split3((xmask >> i) & 1u ? rx[i] : f32x4{0.f, 0.f, 0.f, 0.f}, h, m, l);
        u16* d = sX + hp * kCin + kq * 4;
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + kXPlane) = m;
        *reinterpret_cast<uint2*>(d + 2 * kXPlane) = l;
      }
    }
#pragma unroll
    for (int i = 0; i < kDF4; ++i) {
      const int e = tid + kThreadsH * i, px = e >> 4, cq = e & 15;
      if (px < kBlkPx) {
        uint2 h, m, l;
        split3(rd[i], h, m, l);
        u16* d = sD + px * kCout + ((((cq >> 1) ^ ((px & 2) << 1))) << 3) + (cq & 1) * 4;
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + kDPlane) = m;
        *reinterpret_cast<uint2*>(d + 2 * kDPlane) = l;
      }
    }
  };

  // ---- fragments: transposing reads.  16-lane group G = lane >> 4, Lq = lane & 15: the group reads pixel rows +0..3 (Lq >> 2),
  // channels 16 (G & 1) + 4 (Lq & 3) .. + 3; after the transpose lane (lane & 31) holds 4 consecutive pixels of channel lane & 31;
  // a second read 4 pixel rows further completes the 8 pixels of lane half h = G >> 1 (block row 2 s + h of k-step s).
  const int G = lane >> 4, Lq = lane & 15, fh = G >> 1;
  const int chq = 16 * (G & 1) + 4 * (Lq & 3);               // first channel of this lane's 8-byte piece (within a 32-channel tile)
  auto frag_d = [&](int plane, int s) -> bf16x8 {
    const int row = (2 * s + fh) * kPB + (Lq >> 2);           // pixel of the block; row + 4 shares (row & 2)
    const int ch = mt * 32 + chq;
    const u16* a = sD + plane * kDPlane + row * kCout + ((((ch >> 3) ^ ((row & 2) << 1))) << 3) + (ch & 7);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a + 4 * kCout));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto frag_x = [&](int plane, int s, int tx) -> bf16x8 {
    const int hp = (2 * s + fh + krow) * kHP + tx + (Lq >> 2);          // halo pixel of tap (krow, tx) for block pixel (2 s + fh, Lq >> 2)
    const u16* a = sX + plane * kXPlane + hp * kCin + chq;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a + 4 * kCin));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto compute = [&]() {
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};     // the six terms, smallest first
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 af[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) af[q] = frag_d(q, s);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        bf16x8 bf[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) bf[q] = frag_x(q, s, tx);
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA[t]], bf[TB[t]], acc[tx], 0, 0, 0);
      }
    }
  };

  if (pt_begin < pt_end) {
    gload();
    for (long long pt = pt_begin; pt < pt_end; ++pt) {
      sstore();
      __syncthreads();
      if (pt + 1 < pt_end) gload();
      __builtin_amdgcn_sched_barrier(0);
      compute();
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                              // every wave is done reading this block's planes
    }
  }
  // ---- this workgroup's partial: ws[blk][co][tap * 32 + ci] ----
  float* out = p.ws + (long long)blockIdx.x * kCout * (9 * kCin);
  const int n = lane & 31, r_lane = 4 * (lane >> 5);
#pragma unroll
  for (int tx = 0; tx < 3; ++tx)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + r_lane;
      out[(long long)m * (9 * kCin) + (krow * 3 + tx) * kCin + n] = acc[tx][r];
    }
}

inline int halo_wg_blocks(long long patches) {
  // two workgroups per CU are co-resident (12 of the 16 waves 128 registers allow; LDS would take three): ONE round of 512, every
  // workgroup with at least 8 blocks to amortise its 74 KB partial.  (768 workgroups = 1.5 rounds cost the time of two.)
  long long b = patches / 8;
  if (b > 512) b = 512;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

bool fsd_conv::wgrad3x3_halo_ok(int height, int width, int cin, int cout, int ksize) {
  static const char* env = FSD_TUNE("FSD_WGRAD_HALO");                 // tuning aid: 0 keeps wgrad_kernel
  if (env && env[0] == '0') return false;
  return fsd_conv::f32_split_on() && ksize == 3 && cin == kCin && cout == kCout && height % kPB == 0 && width % kPB == 0;
}

int fsd_conv::wgrad3x3_halo_slots(int batch, int height, int width) {
  return halo_wg_blocks((long long)batch * (height / kPB) * (width / kPB));
}

// -> 0 and *slots_out partials of [64][288] in ws, or an error
int fsd_conv::wgrad3x3_halo(const float* dy, long long dy_ld, const float* x, long long x_ld, float* ws, int batch, int height,
                            int width, int* slots_out, hipStream_t stream) {
  WgradHaloArgs a;
  a.dy = dy; a.x = x; a.ws = ws; a.dy_ld = dy_ld; a.x_ld = x_ld;
  a.B = batch; a.H = height; a.W = width;
  a.bx = width / kPB; a.by = height / kPB;
  a.patches = (long long)batch * a.bx * a.by;
  const int blocks = halo_wg_blocks(a.patches);
  a.per = (int)((a.patches + blocks - 1) / blocks);
  const int used = (int)((a.patches + a.per - 1) / a.per);           // workgroups that have at least one block (<= blocks)
  *slots_out = used;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_halo_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsH);
  if (e != hipSuccess) return (int)e;
  fsd_prof::Scope prof(fsd_prof::kGemmWgrad, 2.0 * (double)batch * height * width * kCout * 9.0 * kCin, stream);
  FSD_LAUNCH(wgrad3x3_halo_kernel, dim3((unsigned)used), dim3(kThreadsH), kLdsH, stream, a);
  return (int)hipGetLastError();
}
