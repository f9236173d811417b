// Shared by the fp32 (conv.hip) and bf16 (conv_bf16.hip) implicit-GEMM kernels: launch arguments, the
// XCD-aware tile mapping and the accumulator epilogue (the C/D layout of the 32x32 MFMAs is dtype-independent).
#ifndef FSD_CONV_COMMON_HPP_
#define FSD_CONV_COMMON_HPP_
#include <hip/hip_runtime.h>
#include <stdlib.h>

namespace fsd_conv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* x;
  const void* w;        // packed weights: float (fp32 path) or bf16 (bf16 path)
  const float* bias;
  float* y;
  float* bn_partial;
  long long x_ld, y_ld;
  int H, W, HW, M;
  int Cout, ks, pad;
  int cpg;       // 16-byte channel groups per tap  (Cin/4)
  int kgroups;   // taps * cpg
  int nk;        // k-chunks
  int cpt;       // FASTK: chunks per tap (Cin/32)
  int Kpad;      // packed weight row length (floats)
  int m_tiles, n_tiles;
  int m_base;     // first output row handled by this launch (tail launches start past the main rows)
  int part_base;  // first bn_partial row of this launch
  // batched GEMMs (gridDim.y > 1, used by the Winograd path): element strides between the operands of consecutive batches
  long long x_bs, w_bs, y_bs;
  int batches;
  float slope;   // leaky slope applied to (acc + bias) in the NHWC epilogue; 1 = linear (fsd_conv2d_fwd_act)
  int wide;      // NHWC epilogue through LDS with float4 stores (needs y 16-byte aligned, y_ld % 4 == 0, Cout % 4 == 0)
  int flat_xcd;  // batched launches: XCD-aware order over the flat (batch, tile) space instead of per batch (conv.hip)
  // activation on load (fsd_conv2d_fwd_ex): the input is the RAW output y of the producing conv and x = leaky(y * in_scale +
  // in_shift) (per input channel) is formed in the staging registers -- the producer's BatchNorm + leaky pass is never
  // run, its result never written or read.  null = the input is used as it is.
  const float* in_scale;
  const float* in_shift;
  float in_slope;
  // split-K of the batched plain GEMMs (conv_gemm_batched, 4-wave split kernel): gridDim.y = batches * ksplit; slice s of batch
  // b reduces k-chunks [s, s + 1) * nk / ksplit into y + b * y_bs + s * y_ks -- the reader (Winograd output transform) adds the
  // slices.  1 = off (every other launch).
  int ksplit = 1;
  long long y_ks = 0;
};

// leaky(v * sc + sh), the expression of bn_act_pool_kernel (elementwise.hip): consumers that apply it on load produce the
// bits the materialised activation would have held
__device__ __forceinline__ f32x4 affine_act4(f32x4 v, f32x4 sc, f32x4 sh, float slope) {
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t = __builtin_fmaf(v[k], sc[k], sh[k]);
    r[k] = t > 0.f ? t : t * slope;
  }
  return r;
}

// fp32 1x1 "convolutions" as a batch of plain GEMMs y[b] = x[b] * w[b]^T (defined in conv.hip)
// ksplit > 1: the reduction is cut into ksplit slices, slice s written at y + s * y_ks (see ConvArgs; batched_ksplit's answer)
int conv_gemm_batched(const float* x, long long x_ld, long long x_bs, const float* w_packed, long long w_bs, float* y,
                      long long y_ld, long long y_bs, long long rows, int cin, int cout, int batches, hipStream_t stream,
                      int ksplit = 1, long long y_ks = 0);
// how many K slices a batched launch of this shape should be cut into (1 = none): few rows per batch (a small batch of images
// or the reweighting net's 3x3 / 7x7 maps) leave the launch with ~2 workgroups per CU, each waiting for one 16 KB chunk at a time
int batched_ksplit(long long rows, int cin, int cout, int batches);

// fp32 -> three bfloat16 planes x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2); the residuals
// are exact: 3 x 8 significant bits = the 24 of fp32) for four values; see conv.hip, conv_gemm_kernel SPLIT.
__device__ __forceinline__ void split3(const f32x4& v, uint2& h, uint2& m, uint2& l) {
  unsigned short hs[4], ms[4], ls[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 b1 = (__bf16)v[i];
    const float r1 = v[i] - (float)b1;
    const __bf16 b2 = (__bf16)r1;
    const float r2 = r1 - (float)b2;
    const __bf16 b3 = (__bf16)r2;
    hs[i] = __builtin_bit_cast(unsigned short, b1);
    ms[i] = __builtin_bit_cast(unsigned short, b2);
    ls[i] = __builtin_bit_cast(unsigned short, b3);
  }
  h = make_uint2((unsigned)hs[0] | ((unsigned)hs[1] << 16), (unsigned)hs[2] | ((unsigned)hs[3] << 16));
  m = make_uint2((unsigned)ms[0] | ((unsigned)ms[1] << 16), (unsigned)ms[2] | ((unsigned)ms[3] << 16));
  l = make_uint2((unsigned)ls[0] | ((unsigned)ls[1] << 16), (unsigned)ls[2] | ((unsigned)ls[3] << 16));
}

// runtime switch of the fp32 GEMM arithmetic (conv.hip): true = six bf16 MFMA terms of three-way split operands
bool f32_split_on();

__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  // consecutive logical tiles on one XCD (own L2): dispatch places block b on XCD b % 8
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


// Epilogue of one workgroup tile.  acc[i][j] are the TM x TN 32x32 accumulators of this wave
// (col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)); for NCHW_OUT the MFMA operands were swapped,
// so rows are output channels and columns are pixels.  `smem` is free to reuse (callers end their main
// loop with a barrier).
template <int BM, int BN, int WAVES_M, int WAVES_N, int TM, int TN, bool NCHW_OUT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[TM][TN], float* smem, int m0, int n0,
                                              int mt, int tid, int lane, int wm, int wn) {
  const int c_lane = lane & 31, r_lane = 4 * (lane >> 5);
  if constexpr (!NCHW_OUT) {
    if (!p.wide) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + c_lane;
        const bool n_ok = n < p.Cout;
        const float bv = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + r_lane;
            if (n_ok && m < p.M) {
              float v = acc[i][j][r] + bv;
              if (p.slope != 1.f) v = v > 0.f ? v : v * p.slope;      // inference: BatchNorm folded into w / bias, leaky here
              p.y[(long long)m * p.y_ld + n] = v;
            }
          }
        }
      }
    }
    if (p.bn_partial != nullptr) {
      // per-tile column sums of the raw outputs (rows past M hold exact zeros)
      float* s_stat = smem;    // [WAVES_M][BN][2]; the main loop's last barrier freed the staging LDS
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r];
            s += v;
            q += v * v;
          }
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (lane < 32) {
          const int col = (wn * TN + j) * 32 + c_lane;
          s_stat[(wm * BN + col) * 2 + 0] = s;
          s_stat[(wm * BN + col) * 2 + 1] = q;
        }
      }
      __syncthreads();
      if (tid < BN) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) {
          s += s_stat[(w * BN + tid) * 2 + 0];
          q += s_stat[(w * BN + tid) * 2 + 1];
        }
        const int n = n0 + tid;
        if (n < p.Cout) {
          float* dst = p.bn_partial + ((long long)(p.part_base + mt) * p.Cout + n) * 2;
          dst[0] = s;
          dst[1] = q;
        }
      }
      if (p.wide) __syncthreads();          // s_stat is read: the tile below re-uses the space
    }
    if (p.wide) {
      // Wide stores.  Straight from the accumulators a lane stores ONE float per instruction (16 x TM x TN instructions per
      // wave, 128 contiguous bytes per pixel row each); store ISSUE, not bandwidth, is what a short-K tile then waits for
      // (MI355X_MICROARCH.md: an epilogue of 16 narrow stores per lane costs ~9 k cycles -- a K = 64 Winograd position tile is
      // 2 k cycles of MFMA).  The tile crosses LDS once ([BM][BN] floats, written as it lies in the accumulators: 32 lanes =
      // 32 consecutive floats, conflict-free) and leaves as float4: 4x fewer store instructions, 256-byte row segments.
      constexpr int NT = WAVES_M * WAVES_N * 64;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nl = (wn * TN + j) * 32 + c_lane;
        const float bv = (p.bias != nullptr && n0 + nl < p.Cout) ? p.bias[n0 + nl] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][j][r] + bv;
            if (p.slope != 1.f) v = v > 0.f ? v : v * p.slope;
            smem[((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + r_lane) * BN + nl] = v;
          }
      }
      __syncthreads();
      constexpr int PPR = BN / 4;           // float4 pieces per tile row
      for (int it = tid; it < BM * PPR; it += NT) {
        const int row = it / PPR, pc = it - row * PPR;
        const int m = m0 + row, n = n0 + pc * 4;
        if (m < p.M && n < p.Cout)          // Cout % 4 == 0 (launcher): whole pieces
          *reinterpret_cast<f32x4*>(p.y + (long long)m * p.y_ld + n) = *reinterpret_cast<const f32x4*>(smem + row * BN + pc * 4);
      }
    }
  } else {
    // transposed accumulator: rows = output channels, cols = pixels -> NCHW store, pixel-contiguous
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + (wm * TM + i) * 32 + c_lane;
      const int b = m / p.HW;
      const int hw = m - b * p.HW;
      const bool m_ok = m < p.M;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + (wn * TN + j) * 32 + (r & 3) + 8 * (r >> 2) + r_lane;
          if (m_ok && n < p.Cout) {
            const float bv = p.bias != nullptr ? p.bias[n] : 0.f;
            p.y[((long long)b * p.Cout + n) * p.HW + hw] = acc[i][j][r] + bv;
          }
        }
      }
    }
  }
}

// halo-staged direct 3x3 convolution of the narrow layers under the split arithmetic (conv_halo.hip)
bool halo_ok(int height, int width, int cin, int cout, int ksize, bool nchw);
// bf16 storage mode: persistent halo-staged 3x3 convolution of the 32 -> 64 / 64 -> 32 layers, weights in registers
// (conv_halo_h.hip); halo_h_rows = BatchNorm partial rows it writes (one per workgroup)
bool halo_h_ok(int height, int width, int cin, int cout, int ksize);
bool halo_h_layout_ok(const void* x, long long x_ld, const void* y, long long y_ld, int batch, int height, int width, int cin,
                      int cout);
int halo_h_rows(int batch, int height, int width, int cin, int cout);
int conv3x3_halo_h(const void* x, long long x_ld, const void* w_packed, int kpad, const float* bias, void* y, long long y_ld,
                   float* bn_partial, int batch, int height, int width, int cin, int cout, float slope, hipStream_t stream);
// bf16 storage mode: weight gradient of the 32 -> 64 / 64 -> 128 3x3 layers on 8 x 16 pixel blocks with a halo patch, one
// partial per persistent workgroup (wgrad_halo_h.hip); slots = partial slices the fold reads
bool wgrad_halo_h_ok(int height, int width, int cin, int cout, int ksize);
int wgrad_halo_h_slots(int batch, int height, int width, int cin);
int wgrad3x3_halo_h(const void* dy, long long dy_ld, const void* x, long long x_ld, float* ws, int batch, int height, int width,
                    int cin, int cout, hipStream_t stream);
// weight gradient of the narrow 3x3 layers on 8 x 8 pixel blocks with a halo patch (wgrad_halo.hip)
bool wgrad3x3_halo_ok(int height, int width, int cin, int cout, int ksize);
int wgrad3x3_halo_slots(int batch, int height, int width);
int wgrad3x3_halo(const float* dy, long long dy_ld, const float* x, long long x_ld, float* ws, int batch, int height, int width,
                  int* slots_out, hipStream_t stream);
int conv3x3_halo(const float* x, long long x_ld, const float* w_packed, int kpad, const float* bias, float* y, long long y_ld,
                 float* bn_partial, int batch, int height, int width, int cin, int cout, float slope, hipStream_t stream);

// which tile the batched forward / data-gradient GEMM of conv_gemm_batched will use; returns the number of M tiles
int conv_gemm_batched_plan(long long rows, int cin, int cout, int* bm_out, int* bn_out, int* dma_out);

// batched reduction GEMMs on the fp32 weight-gradient kernel (defined in wgrad.hip)
int wgrad_batched_splits(long long rows, int cin, int cout, int batches);
// plan of wgrad_gemm_batched: *dma = 128x128 DMA-staged variant, *splits = row splits of the main launch, *tail_rows =
// rows (< 32) handled by the 64x64 side launch; returns the number of workspace slots
int wgrad_batched_plan(long long rows, int cin, int cout, int batches, int* dma, int* splits, int* tail_rows);
int wgrad_gemm_batched(const float* dy, long long dy_ld, long long dy_bs, const float* x, long long x_ld, long long x_bs,
                       float* ws, long long rows, int cin, int cout, int batches, int* splits_out, hipStream_t stream);

}  // namespace fsd_conv
#endif  // FSD_CONV_COMMON_HPP_
