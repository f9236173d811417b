// Region loss (YOLOv2 box/objectness loss + softmax over the N episode classes), fused
// forward + gradient, for gfx950.  Replaces the reference's device->host copy + python
// target-assignment loops (region_loss.py:37-132, 252-366) by three small HBM-bound kernels.
//
// Compiled with -ffp-contract=off: the float32 IoU "silence" test and the float64 anchor
// matching must round exactly like the reference's separate tensor / python-float operations.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fsdet.h"
#include "profile.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxGT = 50;          // region_loss.py:85 hard-codes 50 boxes per (image, class) row
constexpr int kMaxAnchors = 16;

struct RegionArgs {
  const float* out;        // (rows, A*(5+C), H, W)
  const double* target;    // (rows, L)  [cls, cx, cy, w, h] x 50, zero-terminated on cx
  const int* keep;         // (rows) compact index of a kept row, or -1 if neg_filter dropped it
  float* grad;             // same shape as out
  double* stats;           // 16 doubles, see fsdet.h
  int* img_count;          // (images, A, H, W) how many kept rows claim the cell
  float* img_tcls;         // (images, A, H, W) sum of their class targets
  float* dbg;              // optional 9 x (kept, A, H, W): coord,conf,cls masks, tx,ty,tw,th,tconf,tcls
  const float* pred;       // kFromPred only: decoded boxes (rows*A*H*W, 4) = [x, y, w, h] in grid cells
  long long dbg_stride;
  int rows, rows_per_image, A, C, H, W, L, max_boxes;
  int early;               // seen < 12800
  int zero_tcls;           // v1 with cfg.metayolo: class targets forced to 0
  float coord_scale, noobject_scale, object_scale, class_scale, thresh;
  double aw[kMaxAnchors], ah[kMaxAnchors];
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// utils.bbox_iou, centre format, python-double arithmetic (utils.py:21-52)
__device__ double iou_f64(double x1, double y1, double w1, double h1,
                          double x2, double y2, double w2, double h2) {
  double mx = fmin(x1 - w1 / 2.0, x2 - w2 / 2.0);
  double Mx = fmax(x1 + w1 / 2.0, x2 + w2 / 2.0);
  double my = fmin(y1 - h1 / 2.0, y2 - h2 / 2.0);
  double My = fmax(y1 + h1 / 2.0, y2 + h2 / 2.0);
  double cw = w1 + w2 - (Mx - mx);
  double ch = h1 + h2 - (My - my);
  if (cw <= 0 || ch <= 0) return 0.0;
  double carea = cw * ch;
  return carea / (w1 * h1 + w2 * h2 - carea);
}

// utils.bbox_ious, float32 tensor arithmetic in the reference's operation order (utils.py:54-83)
__device__ __forceinline__ float iou_f32(float px, float py, float pw, float ph,
                                         float gx, float gy, float gw, float gh) {
  float mx = fminf(px - pw / 2.0f, gx - gw / 2.0f);
  float Mx = fmaxf(px + pw / 2.0f, gx + gw / 2.0f);
  float my = fminf(py - ph / 2.0f, gy - gh / 2.0f);
  float My = fmaxf(py + ph / 2.0f, gy + gh / 2.0f);
  float cw = pw + gw - (Mx - mx);
  float ch = ph + gh - (My - my);
  float carea = (cw <= 0.0f || ch <= 0.0f) ? 0.0f : cw * ch;
  float uarea = pw * ph + gw * gh - carea;
  return carea / uarea;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// One workgroup per (image, class) row.  kFromPred: the standalone build_targets (region_loss.py:37-132) -- boxes come
// decoded from the caller, only the nine target tensors and nGT / nCorrect are produced (no loss, no gradient).
template <bool kFromPred>
__global__ __launch_bounds__(kThreads) void region_rows_kernel(RegionArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int HW = p.H * p.W;
  const int cells = p.A * HW;
  const int chans = 5 + p.C;
  double* s_gt = reinterpret_cast<double*>(smem);                  // kMaxGT * 5
  double* s_red = s_gt + kMaxGT * 5;                               // 9 * (kThreads/64)  (doubles first: 8-byte aligned)
  int* s_cnt = reinterpret_cast<int*>(s_red + 9 * (kThreads / 64)); // [0]=#gt for assignment, [1]=#gt for silence
  int* s_owner = s_cnt + 4;                                        // cells

  const int row = blockIdx.x;
  const int tid = threadIdx.x;
  const long long row_off = (long long)row * p.A * chans * HW;
  const float* o = kFromPred ? nullptr : p.out + row_off;
  float* g = kFromPred ? nullptr : p.grad + row_off;
  const int kept = kFromPred ? row : p.keep[row];
  const float* pr = kFromPred ? p.pred + (long long)row * cells * 4 : nullptr;

  if (!kFromPred && kept < 0) {   // dropped by neg_filter: no box/objectness gradient
    for (int c = tid; c < cells; c += kThreads) {
      int a = c / HW, hw = c - a * HW;
      float* ga = g + (long long)a * chans * HW + hw;
#pragma unroll
      for (int k = 0; k < 5; ++k) ga[(long long)k * HW] = 0.0f;
    }
    return;
  }

  const double* trow = p.target + (long long)row * p.L;
  for (int i = tid; i < kMaxGT * 5; i += kThreads) s_gt[i] = (i < p.L) ? trow[i] : 0.0;
  for (int c = tid; c < cells; c += kThreads) s_owner[c] = -1;
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    while (n < kMaxGT && n * 5 + 1 < p.L && s_gt[n * 5 + 1] != 0.0) ++n;
    s_cnt[0] = n;
    s_cnt[1] = n < p.max_boxes ? n : p.max_boxes;
  }
  __syncthreads();
  const int n_gt = s_cnt[0];
  const int n_sil = s_cnt[1];

  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;

  // ---- one responsible (anchor, cell) per ground truth; the latest box wins a contested cell
  if (tid < n_gt) {
    const double gx = s_gt[tid * 5 + 1] * p.W, gy = s_gt[tid * 5 + 2] * p.H;
    const double gw = s_gt[tid * 5 + 3] * p.W, gh = s_gt[tid * 5 + 4] * p.H;
    const int gi = (int)gx, gj = (int)gy;
    double best = 0.0;
    int best_n = -1;
    for (int n = 0; n < p.A; ++n) {
      double v = iou_f64(0.0, 0.0, p.aw[n], p.ah[n], 0.0, 0.0, gw, gh);
      if (v > best) { best = v; best_n = n; }
    }
    if (best_n < 0 || gi < 0 || gi >= p.W || gj < 0 || gj >= p.H) {
      atomicAdd(&p.stats[FSD_STAT_BAD_TARGET], 1.0);   // the reference would raise here
    } else {
      const int cell = best_n * HW + gj * p.W + gi;
      atomicMax(&s_owner[cell], tid);
      double px, py, pw, ph;
      if (kFromPred) {
        const float* pc = pr + (long long)cell * 4;
        px = (double)pc[0]; py = (double)pc[1]; pw = (double)pc[2]; ph = (double)pc[3];
      } else {
        const float* oc = o + (long long)best_n * chans * HW + gj * p.W + gi;
        px = (double)(sigmoidf_(oc[0]) + (float)gi);
        py = (double)(sigmoidf_(oc[HW]) + (float)gj);
        pw = (double)(expf(oc[2 * HW]) * (float)p.aw[best_n]);
        ph = (double)(expf(oc[3 * HW]) * (float)p.ah[best_n]);
      }
      if (iou_f64(gx, gy, gw, gh, px, py, pw, ph) > 0.5) acc[7] += 1.0;
    }
    acc[6] += 1.0;
  }
  __syncthreads();

  const float sq_noobj = sqrtf(p.noobject_scale), sq_obj = sqrtf(p.object_scale);
  float* dbg = p.dbg ? p.dbg + (long long)kept * cells : nullptr;
  const int img = row / p.rows_per_image;

  for (int c = tid; c < cells; c += kThreads) {
    const int a = c / HW, hw = c - a * HW;
    const int j = hw / p.W, i = hw - j * p.W;
    float ow = 0.f, oh = 0.f, x = 0.f, y = 0.f, conf = 0.f, px, py, pw, ph;
    float* gc = nullptr;
    if (kFromPred) {
      const float* pc = pr + (long long)c * 4;
      px = pc[0]; py = pc[1]; pw = pc[2]; ph = pc[3];
    } else {
      const float* oc = o + (long long)a * chans * HW + hw;
      gc = g + (long long)a * chans * HW + hw;
      ow = oc[2 * HW]; oh = oc[3 * HW];
      x = sigmoidf_(oc[0]); y = sigmoidf_(oc[HW]); conf = sigmoidf_(oc[4 * HW]);
      px = x + (float)i; py = y + (float)j;
      pw = expf(ow) * (float)p.aw[a]; ph = expf(oh) * (float)p.ah[a];
    }

    float best = 0.0f;
    for (int t = 0; t < n_sil; ++t) {
      const float gx = (float)(s_gt[t * 5 + 1] * p.W), gy = (float)(s_gt[t * 5 + 2] * p.H);
      const float gw = (float)(s_gt[t * 5 + 3] * p.W), gh = (float)(s_gt[t * 5 + 4] * p.H);
      const float v = iou_f32(px, py, pw, ph, gx, gy, gw, gh);
      best = (v > best || v != v) ? v : best;           // torch.max keeps NaN
    }
    float conf_mask = (best > p.thresh) ? 0.0f : p.noobject_scale;
    float sq = (best > p.thresh) ? 0.0f : sq_noobj;
    float coord_mask = p.early ? 1.0f : 0.0f;
    float tx = p.early ? 0.5f : 0.0f, ty = tx, tw = 0.0f, th = 0.0f, tconf = 0.0f, tcls = 0.0f, cls_mask = 0.0f;

    const int t = s_owner[c];
    if (t >= 0) {
      const double gx = s_gt[t * 5 + 1] * p.W, gy = s_gt[t * 5 + 2] * p.H;
      const double gw = s_gt[t * 5 + 3] * p.W, gh = s_gt[t * 5 + 4] * p.H;
      coord_mask = 1.0f; cls_mask = 1.0f; conf_mask = p.object_scale; sq = sq_obj;
      tx = (float)(gx - (double)(int)gx);
      ty = (float)(gy - (double)(int)gy);
      tw = (float)log(gw / p.aw[a]);
      th = (float)log(gh / p.ah[a]);
      tconf = (float)iou_f64(gx, gy, gw, gh, (double)px, (double)py, (double)pw, (double)ph);
      tcls = p.zero_tcls ? 0.0f : (float)s_gt[t * 5];
      if (!kFromPred) {
        atomicAdd(&p.img_count[(long long)img * cells + c], 1);
        atomicAdd(&p.img_tcls[(long long)img * cells + c], tcls);
      }
    }

    if (!kFromPred) {
    // 0.5 * sum((pred*mask - target*mask)^2), written like the reference's MSELoss operands
    const float dx = x * coord_mask - tx * coord_mask;
    const float dy = y * coord_mask - ty * coord_mask;
    const float dw = ow * coord_mask - tw * coord_mask;
    const float dh = oh * coord_mask - th * coord_mask;
    const float dc = conf * sq - tconf * sq;
    acc[0] += (double)(dx * dx); acc[1] += (double)(dy * dy);
    acc[2] += (double)(dw * dw); acc[3] += (double)(dh * dh);
    acc[4] += (double)(dc * dc);
    if (conf > 0.25f) acc[8] += 1.0;

    gc[0] = p.coord_scale * dx * coord_mask * (x * (1.0f - x));
    gc[HW] = p.coord_scale * dy * coord_mask * (y * (1.0f - y));
    gc[2 * HW] = p.coord_scale * dw * coord_mask;
    gc[3 * HW] = p.coord_scale * dh * coord_mask;
    gc[4 * HW] = dc * sq * (conf * (1.0f - conf));
    }

    if (dbg) {
      dbg[c] = coord_mask;                    dbg[p.dbg_stride + c] = conf_mask;
      dbg[2 * p.dbg_stride + c] = cls_mask;   dbg[3 * p.dbg_stride + c] = tx;
      dbg[4 * p.dbg_stride + c] = ty;         dbg[5 * p.dbg_stride + c] = tw;
      dbg[6 * p.dbg_stride + c] = th;         dbg[7 * p.dbg_stride + c] = tconf;
      dbg[8 * p.dbg_stride + c] = tcls;
    }
  }

  // ---- block reduction -> one double atomic per statistic per row
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    double v = wave_sum(acc[k]);
    if (lane == 0) s_red[k * (kThreads / 64) + wave] = v;
  }
  __syncthreads();
  if (tid < 9) {
    double v = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) v += s_red[tid * (kThreads / 64) + w];
    if (tid < 4) v *= 0.5 * (double)p.coord_scale;
    if (tid == 4) v *= 0.5;
    if (v != 0.0) atomicAdd(&p.stats[tid], v);
  }
}

// Class term.  One thread per (image, anchor, cell) group of `n_logits` logits spaced `stride`
// floats apart:  v2 -> the N rows of the image (softmax across episode classes),
//                v1 -> the C class channels of the cell.
__global__ __launch_bounds__(kThreads) void region_class_kernel(RegionArgs p, int groups, int n_logits,
                                                                long long stride) {
  const int HW = p.H * p.W;
  const int cells = p.A * HW;
  const int chans = 5 + p.C;
  const int gid = blockIdx.x * kThreads + threadIdx.x;
  double loss = 0.0;
  if (gid < groups) {
    const int img = gid / cells;             // v2: image;  v1: row
    const int c = gid - img * cells;
    const int a = c / HW, hw = c - a * HW;
    const long long base = (long long)img * p.rows_per_image * p.A * chans * HW +
                           ((long long)a * chans + 5) * HW + hw;
    const float* lo = p.out + base;
    float* gr = p.grad + base;
    const bool on = p.img_count[gid] == 1;
    if (!on) {
      for (int n = 0; n < n_logits; ++n) gr[n * stride] = 0.0f;
    } else {
      int label = (int)p.img_tcls[gid];
      if (label < 0 || label >= n_logits) {
        atomicAdd(&p.stats[FSD_STAT_BAD_TARGET], 1.0);
        label = 0;
      }
      float m = -INFINITY;
      for (int n = 0; n < n_logits; ++n) m = fmaxf(m, lo[n * stride]);
      float s = 0.0f;
      for (int n = 0; n < n_logits; ++n) s += expf(lo[n * stride] - m);
      const float lse = m + logf(s);
      loss = (double)(p.class_scale * (lse - lo[label * stride]));
      const float inv = 1.0f / s;
      for (int n = 0; n < n_logits; ++n) {
        const float sm = expf(lo[n * stride] - m) * inv;
        gr[n * stride] = p.class_scale * (sm - (n == label ? 1.0f : 0.0f));
      }
    }
  }
  __shared__ double s_red[kThreads / 64];
  double v = wave_sum(loss);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) t += s_red[w];
    if (t != 0.0) atomicAdd(&p.stats[5], t);
  }
}

__global__ void region_finalize_kernel(const double* stats, float* loss_out) {
  float t = 0.0f;
#pragma unroll
  for (int k = 0; k < 6; ++k) t += (float)stats[k];
  loss_out[0] = t;
}

// Rows dropped by neg_filter keep zero class gradient in v1 (their logits never reach the loss)
}  // namespace

extern "C" size_t fsd_region_loss_workspace_bytes(int rows, int rows_per_image, int num_anchors,
                                                  int height, int width) {
  size_t images = (size_t)(rows / (rows_per_image > 0 ? rows_per_image : 1));
  size_t cells = (size_t)num_anchors * height * width;
  return FSD_REGION_STATS * sizeof(double) + images * cells * (sizeof(int) + sizeof(float));
}

extern "C" int fsd_region_loss_fwd_bwd(const float* output, const double* target, const int* keep,
                                       float* grad_output, float* loss_out, void* workspace,
                                       size_t workspace_bytes, int rows, int rows_per_image,
                                       int num_anchors, int num_classes, int height, int width,
                                       int target_len, const double* anchors_host,
                                       float coord_scale, float noobject_scale, float object_scale,
                                       float class_scale, float thresh, long long seen,
                                       int max_boxes, int softmax_over_rows, int zero_tcls,
                                       float* dbg_targets, hipStream_t stream) {
  (void)hipGetLastError();   // drop a stale error left by someone else's earlier call
  if (!output || !target || !keep || !grad_output || !loss_out || !workspace || !anchors_host)
    return FSD_ERR_ARG;
  if (num_anchors < 1 || num_anchors > kMaxAnchors || rows_per_image < 1 || rows % rows_per_image)
    return FSD_ERR_ARG;
  if (softmax_over_rows && num_classes != 1) return FSD_ERR_UNSUPPORTED;
  if (workspace_bytes < fsd_region_loss_workspace_bytes(rows, rows_per_image, num_anchors, height, width))
    return FSD_ERR_WORKSPACE;
  if (rows == 0) return FSD_ERR_ARG;

  RegionArgs p;
  p.out = output; p.target = target; p.keep = keep; p.grad = grad_output;
  const int images = rows / rows_per_image;
  const int cells = num_anchors * height * width;
  p.stats = reinterpret_cast<double*>(workspace);
  p.img_count = reinterpret_cast<int*>(p.stats + FSD_REGION_STATS);
  p.img_tcls = reinterpret_cast<float*>(p.img_count + (size_t)images * cells);
  p.dbg = dbg_targets;
  p.rows = rows; p.rows_per_image = rows_per_image; p.A = num_anchors; p.C = num_classes;
  p.H = height; p.W = width; p.L = target_len;
  p.max_boxes = max_boxes < kMaxGT ? max_boxes : kMaxGT;
  p.early = seen < 12800 ? 1 : 0;
  p.zero_tcls = zero_tcls;
  p.coord_scale = coord_scale; p.noobject_scale = noobject_scale; p.object_scale = object_scale;
  p.class_scale = class_scale; p.thresh = thresh;
  for (int a = 0; a < num_anchors; ++a) { p.aw[a] = anchors_host[2 * a]; p.ah[a] = anchors_host[2 * a + 1]; }
  // dbg planes are (kept_rows, A, H, W); the caller sizes them for `rows` rows
  p.dbg_stride = (long long)rows * cells;

  hipError_t e = hipMemsetAsync(workspace, 0,
                                fsd_region_loss_workspace_bytes(rows, rows_per_image, num_anchors, height, width),
                                stream);
  if (e != hipSuccess) return (int)e;
  size_t lds = kMaxGT * 5 * sizeof(double) + (size_t)cells * sizeof(int) + 4 * sizeof(int) +
               9 * (kThreads / 64) * sizeof(double);
  lds = (lds + 15) & ~(size_t)15;
  if (lds > 64 * 1024) return FSD_ERR_UNSUPPORTED;
  // algorithmic bytes of the loss: head output read once, its gradient written once, the float64 targets read once
  const double io_bytes = 2.0 * 4.0 * rows * (double)num_anchors * (5 + num_classes) * height * width + 8.0 * rows * (double)target_len;
  fsd_prof::Scope prof(fsd_prof::kRegion, io_bytes, stream);
  FSD_LAUNCH(region_rows_kernel<false>, dim3(rows), dim3(kThreads), lds, stream, p);

  int groups, n_logits;
  long long stride;
  if (softmax_over_rows) {          // RegionLossV2: softmax across the N class rows of an image
    groups = images * cells; n_logits = rows_per_image;
    stride = (long long)num_anchors * (5 + num_classes) * height * width;
  } else {                          // RegionLoss v1: per-cell softmax over C class channels
    groups = rows * cells; n_logits = num_classes;
    stride = (long long)height * width;
  }
  FSD_LAUNCH(region_class_kernel, dim3((groups + kThreads - 1) / kThreads), dim3(kThreads), 0,
                     stream, p, groups, n_logits, stride);
  FSD_LAUNCH(region_finalize_kernel, dim3(1), dim3(1), 0, stream, p.stats, loss_out);
  return (int)hipGetLastError();
}

extern "C" int fsd_region_build_targets(const float* pred_boxes, const double* target, float* targets_out,
                                        double* stats, int rows, int num_anchors, int height, int width,
                                        int target_len, const double* anchors_host, float noobject_scale,
                                        float object_scale, float thresh, long long seen, int max_boxes,
                                        hipStream_t stream) {
  (void)hipGetLastError();
  if (!pred_boxes || !target || !targets_out || !stats || !anchors_host) return FSD_ERR_ARG;
  if (num_anchors < 1 || num_anchors > kMaxAnchors || rows < 1) return FSD_ERR_ARG;
  RegionArgs p = {};
  p.pred = pred_boxes; p.target = target; p.dbg = targets_out; p.stats = stats;
  const int cells = num_anchors * height * width;
  p.rows = rows; p.rows_per_image = 1; p.A = num_anchors; p.C = 0; p.H = height; p.W = width; p.L = target_len;
  p.max_boxes = max_boxes < kMaxGT ? max_boxes : kMaxGT;
  p.early = seen < 12800 ? 1 : 0;
  p.coord_scale = 1.f; p.noobject_scale = noobject_scale; p.object_scale = object_scale; p.class_scale = 1.f;
  p.thresh = thresh;
  for (int a = 0; a < num_anchors; ++a) { p.aw[a] = anchors_host[2 * a]; p.ah[a] = anchors_host[2 * a + 1]; }
  p.dbg_stride = (long long)rows * cells;
  hipError_t e = hipMemsetAsync(stats, 0, FSD_REGION_STATS * sizeof(double), stream);
  if (e != hipSuccess) return (int)e;
  size_t lds = kMaxGT * 5 * sizeof(double) + (size_t)cells * sizeof(int) + 4 * sizeof(int) +
               9 * (kThreads / 64) * sizeof(double);
  lds = (lds + 15) & ~(size_t)15;
  if (lds > 64 * 1024) return FSD_ERR_UNSUPPORTED;
  FSD_LAUNCH(region_rows_kernel<true>, dim3(rows), dim3(kThreads), lds, stream, p);
  return (int)hipGetLastError();
}

// ---- inference-side decode (utils.get_region_boxes_v2, utils.py:195-290) ------------------------
namespace {

struct DecodeArgs {
  const float* out;     // (rows, A*(5+C), H, W)
  float* boxes;         // [rows][cap][8]: key, cx/W, cy/H, w/W, h/H, det_conf, cls_conf, cls_id
  int* counts;          // [rows]
  int rows, rows_per_image, A, C, H, W, cap, only_objectness, softmax_over_rows;
  float thresh;
  float aw[kMaxAnchors], ah[kMaxAnchors];
};

// One thread per (row, anchor, cell).  Survivors are appended with an atomic per-row cursor; `key`
// = (cy*W + cx)*A + a is the reference's visiting order so the host can restore it with one sort.
__global__ __launch_bounds__(kThreads) void region_decode_kernel(DecodeArgs p) {
  const int HW = p.H * p.W;
  const int cells = p.A * HW;
  const int chans = 5 + p.C;
  const long long gid = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long long)p.rows * cells) return;
  const int row = (int)(gid / cells);
  const int c = (int)(gid - (long long)row * cells);
  const int a = c / HW, hw = c - a * HW;
  const int j = hw / p.W, i = hw - j * p.W;
  const long long row_stride = (long long)p.A * chans * HW;
  const float* oc = p.out + row * row_stride + (long long)a * chans * HW + hw;
  const float det = sigmoidf_(oc[4 * HW]);
  float cls_conf, cls_id = 0.f;
  if (p.softmax_over_rows) {         // class confidence = softmax across the N rows of this image
    const int img = row / p.rows_per_image;
    const float* base = p.out + (long long)img * p.rows_per_image * row_stride + ((long long)a * chans + 5) * HW + hw;
    float m = -INFINITY;
    for (int n = 0; n < p.rows_per_image; ++n) m = fmaxf(m, base[n * row_stride]);
    float s = 0.f;
    for (int n = 0; n < p.rows_per_image; ++n) s += expf(base[n * row_stride] - m);
    cls_conf = expf(oc[5 * HW] - m) / s;
  } else {                           // classic per-cell softmax over the C class channels
    float m = -INFINITY;
    for (int k = 0; k < p.C; ++k) m = fmaxf(m, oc[(5 + k) * HW]);
    float s = 0.f, best = -1.f;
    for (int k = 0; k < p.C; ++k) s += expf(oc[(5 + k) * HW] - m);
    for (int k = 0; k < p.C; ++k) {
      const float v = expf(oc[(5 + k) * HW] - m) / s;
      if (v > best) { best = v; cls_id = (float)k; }
    }
    cls_conf = best;
  }
  const float conf = p.only_objectness ? det : det * cls_conf;
  if (!(conf > p.thresh)) return;
  const int slot = atomicAdd(&p.counts[row], 1);
  if (slot >= p.cap) return;
  float* b = p.boxes + ((long long)row * p.cap + slot) * 8;
  b[0] = (float)((j * p.W + i) * p.A + a);
  b[1] = (sigmoidf_(oc[0]) + (float)i) / (float)p.W;
  b[2] = (sigmoidf_(oc[HW]) + (float)j) / (float)p.H;
  b[3] = expf(oc[2 * HW]) * p.aw[a] / (float)p.W;
  b[4] = expf(oc[3 * HW]) * p.ah[a] / (float)p.H;
  b[5] = det;
  b[6] = cls_conf;
  b[7] = cls_id;
}

}  // namespace

extern "C" int fsd_region_decode(const float* output, float* boxes, int* counts, int rows, int rows_per_image,
                                 int num_anchors, int num_classes, int height, int width, const double* anchors_host,
                                 float conf_thresh, int only_objectness, int softmax_over_rows, int cap,
                                 hipStream_t stream) {
  (void)hipGetLastError();
  if (!output || !boxes || !counts || !anchors_host || rows < 1 || cap < 1) return FSD_ERR_ARG;
  if (num_anchors < 1 || num_anchors > kMaxAnchors || rows_per_image < 1 || rows % rows_per_image) return FSD_ERR_ARG;
  if (softmax_over_rows && num_classes != 1) return FSD_ERR_UNSUPPORTED;
  DecodeArgs p;
  p.out = output; p.boxes = boxes; p.counts = counts; p.rows = rows; p.rows_per_image = rows_per_image;
  p.A = num_anchors; p.C = num_classes; p.H = height; p.W = width; p.cap = cap;
  p.only_objectness = only_objectness; p.softmax_over_rows = softmax_over_rows; p.thresh = conf_thresh;
  for (int a = 0; a < num_anchors; ++a) { p.aw[a] = (float)anchors_host[2 * a]; p.ah[a] = (float)anchors_host[2 * a + 1]; }
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * rows, stream);
  if (e != hipSuccess) return (int)e;
  const long long total = (long long)rows * num_anchors * height * width;
  FSD_LAUNCH(region_decode_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, p);
  return (int)hipGetLastError();
}

// ---- greedy non-maximum suppression (utils.nms, utils.py:85-104) --------------------------------
namespace {

constexpr int kNmsMax = 2048;       // boxes per (image, class) row held in LDS (19x19x5 = 1805 at 608x608)

// One workgroup per (image, class) row of fsd_region_decode's output.  The reference sorts the row by the FLOAT32
// value 1 - det_conf (a FloatTensor, utils.py:89-93) -- distinct confidences can share a key, and ties keep the
// visiting order -- then walks it: a box that is still alive suppresses (det_conf := 0) every later box whose
// python-double IoU with it exceeds the threshold.  Here: 64-bit composite keys {fp32 bits of 1-det, visiting order,
// slot} sorted with a bitonic network in LDS, then the same walk with the inner loop spread over the workgroup.
__global__ __launch_bounds__(kThreads) void region_nms_kernel(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                               int cap, float thresh, int* __restrict__ keep_idx,
                                                               int* __restrict__ keep_counts) {
  __shared__ unsigned long long s_key[kNmsMax];
  __shared__ float s_box[kNmsMax][5];       // cx, cy, w, h, det in sorted order
  __shared__ int s_alive[kNmsMax];
  const int row = blockIdx.x, tid = threadIdx.x;
  int n = counts[row];
  n = n < cap ? n : cap;
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  const float* rb = boxes + (long long)row * cap * 8;
  for (int i = tid; i < np2; i += kThreads) {
    unsigned long long k = ~0ull;           // padding sorts last
    if (i < n) {
      const float key = 1.0f - rb[i * 8 + 5];
      k = ((unsigned long long)__float_as_uint(key) << 32) | ((unsigned long long)(unsigned)rb[i * 8] << 16) | (unsigned)i;
    }
    s_key[i] = k;
  }
  __syncthreads();
  for (int size = 2; size <= np2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < np2 / 2; t += kThreads) {
        const int lo = 2 * t - (t & (stride - 1));          // index with bit `stride` clear
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = s_key[lo], b = s_key[hi];
        if ((a > b) == up) { s_key[lo] = b; s_key[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n; i += kThreads) {
    const int slot = (int)(s_key[i] & 0xffffu);
    const float* b = rb + slot * 8;
    s_box[i][0] = b[1]; s_box[i][1] = b[2]; s_box[i][2] = b[3]; s_box[i][3] = b[4]; s_box[i][4] = b[5];
    s_alive[i] = b[5] > 0.f ? 1 : 0;
  }
  __syncthreads();
  const double th = (double)thresh;
  for (int i = 0; i < n; ++i) {
    if (s_alive[i]) {                        // uniform: every thread reads the same LDS word after the barrier
      const double x1 = s_box[i][0], y1 = s_box[i][1], w1 = s_box[i][2], h1 = s_box[i][3];
      for (int j = i + 1 + tid; j < n; j += kThreads)
        if (s_alive[j] && iou_f64(x1, y1, w1, h1, s_box[j][0], s_box[j][1], s_box[j][2], s_box[j][3]) > th) s_alive[j] = 0;
      __syncthreads();
    }
  }
  // compaction of the survivors in sorted order (wave 0, ballot prefix)
  if (tid < 64) {
    int off = 0;
    for (int base = 0; base < n; base += 64) {
      const int i = base + tid;
      const bool flag = i < n && s_alive[i];
      const unsigned long long m = __ballot(flag);
      if (flag) keep_idx[(long long)row * cap + off + __popcll(m & ((1ull << tid) - 1ull))] = (int)(s_key[i] & 0xffffu);
      off += __popcll(m);
    }
    if (tid == 0) keep_counts[row] = off;
  }
}

}  // namespace

extern "C" int fsd_region_nms(const float* boxes, const int* counts, int rows, int cap, float nms_thresh, int* keep_idx,
                              int* keep_counts, hipStream_t stream) {
  (void)hipGetLastError();
  if (!boxes || !counts || !keep_idx || !keep_counts || rows < 1 || cap < 1) return FSD_ERR_ARG;
  if (cap > kNmsMax) return FSD_ERR_UNSUPPORTED;          // also keeps slot and visiting order inside 16 bits
  FSD_LAUNCH(region_nms_kernel, dim3(rows), dim3(kThreads), 0, stream, boxes, counts, cap, nms_thresh, keep_idx,
                     keep_counts);
  return (int)hipGetLastError();
}
