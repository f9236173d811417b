// Episode input pipeline on the device: the image half of the reference's CPU-side loader,
// image.data_augmentation + torchvision ToTensor (image.py:13-87, dataset.py:240-263, train_meta.py:176-178), as one
// HBM-bound gather kernel that writes the network input directly (NCHW fp32 for the module API, or 16-byte NHWC4 pixels
// -- RGB + support mask -- which the first-layer kernels consume without a layout pass).
//
// Per output pixel: jitter crop + nearest resize + horizontal flip are one table lookup per axis (the host builds the
// two index tables with Pillow's own accumulated-double arithmetic, episode.index_tables), then the colour distortion:
// RGB -> HSV (uint8), three 256-entry tables (hue shift with wrap, saturation / exposure scale with C-int truncation),
// HSV -> RGB, /255.  The conversions restate Pillow's Convert.c (rgb2hsv_row / hsv2rgb) INCLUDING which sub-expressions
// C evaluates in float and which in double; compiled with -ffp-contract=off so no product is fused into an fma.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fsdet.h"
#include "profile.hpp"

namespace {

struct AugArgs {
  const uint8_t* src;        // packed source images, uint8 RGB, HWC
  const long long* img_off;  // [B] byte offset of image b
  const int* img_w;          // [B] source width (row pitch = 3 * width)
  const int* xtab;           // [B][out_w] source column per output column, -1 = outside the image (black)
  const int* ytab;           // [B][out_h]
  const uint8_t* luts;       // [B][3][256] H, S, V tables, or null: no colour distortion
  const int* distort;        // [B] 0 = skip the colour distortion for this image (its tables are ignored), or null = all on
  const int* mask_box;       // [B][4] x1, y1, x2, y2 of the support mask rectangle (NHWC4 channel 3), or null
  float* out;
  int out_h, out_w, layout;
  long long total;
};

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

__device__ __forceinline__ void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) { uh = 0; us = 0; return; }
  const float cr = (float)(maxc - minc);
  const float s = cr / (float)maxc;
  const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
  float h;
  if (r == maxc) h = bc - gc;                                   // float - float
  else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);   // the 2.0 / 4.0 literals make these double expressions
  else h = (float)(4.0 + (double)gc - (double)rc);
  h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
  uh = clip8((int)((double)h * 255.0));
  us = clip8((int)((double)s * 255.0));
}

__device__ __forceinline__ void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) { r = g = b = v; return; }
  const double h6 = (double)(float)h * 6.0 / 255.0;
  const int i = (int)floor(h6);
  const double f = h6 - (double)i;
  const double fs = (double)(float)s / 255.0;
  const int p = clip8((int)round((double)(float)v * (1.0 - fs)));
  const int q = clip8((int)round((double)(float)v * (1.0 - fs * f)));
  const int t = clip8((int)round((double)(float)v * (1.0 - fs * (1.0 - f))));
  switch (i % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

__global__ __launch_bounds__(256) void augment_kernel(AugArgs a) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.total) return;
  const int x = (int)(idx % a.out_w);
  const long long t = idx / a.out_w;
  const int y = (int)(t % a.out_h);
  const int bimg = (int)(t / a.out_h);
  const int sx = a.xtab[(long long)bimg * a.out_w + x], sy = a.ytab[(long long)bimg * a.out_h + y];
  int r = 0, g = 0, b = 0;
  if (sx >= 0 && sy >= 0) {
    const uint8_t* px = a.src + a.img_off[bimg] + ((long long)sy * a.img_w[bimg] + sx) * 3;
    r = px[0]; g = px[1]; b = px[2];
  }
  if (a.luts != nullptr && (a.distort == nullptr || a.distort[bimg] != 0)) {
    const uint8_t* l = a.luts + (long long)bimg * 768;
    int h, s, v;
    rgb2hsv(r, g, b, h, s, v);
    hsv2rgb(l[h], l[256 + s], l[512 + v], r, g, b);
  }
  const float fr = (float)r / 255.0f, fg = (float)g / 255.0f, fb = (float)b / 255.0f;     // ToTensor
  if (a.layout == 1) {
    float m = 0.f;
    if (a.mask_box != nullptr) {
      const int* mb = a.mask_box + bimg * 4;
      m = (x >= mb[0] && x < mb[2] && y >= mb[1] && y < mb[3]) ? 1.f : 0.f;
    }
    reinterpret_cast<float4*>(a.out)[idx] = make_float4(fr, fg, fb, m);
  } else {
    const long long plane = (long long)a.out_h * a.out_w;
    float* o = a.out + (long long)bimg * 3 * plane + (long long)y * a.out_w + x;
    o[0] = fr; o[plane] = fg; o[2 * plane] = fb;
  }
}

}  // namespace

extern "C" int fsd_augment_batch(const unsigned char* src, const long long* img_off, const int* img_w, const int* xtab,
                                 const int* ytab, const unsigned char* luts, const int* distort, const int* mask_box, float* out, int batch,
                                 int out_h, int out_w, int layout, hipStream_t stream) {
  (void)hipGetLastError();
  if (!src || !img_off || !img_w || !xtab || !ytab || !out || batch < 1 || out_h < 1 || out_w < 1) return FSD_ERR_ARG;
  if (layout != 0 && layout != 1) return FSD_ERR_ARG;
  if (layout == 0 && mask_box) return FSD_ERR_UNSUPPORTED;        // the mask is channel 3 of the NHWC4 layout
  if (layout == 1 && (reinterpret_cast<uintptr_t>(out) & 15)) return FSD_ERR_ARG;
  AugArgs a;
  a.src = src; a.img_off = img_off; a.img_w = img_w; a.xtab = xtab; a.ytab = ytab; a.luts = luts; a.distort = distort; a.mask_box = mask_box;
  a.out = out; a.out_h = out_h; a.out_w = out_w; a.layout = layout;
  a.total = (long long)batch * out_h * out_w;
  const long long blocks = (a.total + 255) / 256;
  if (blocks > 0x7fffffffLL) return FSD_ERR_UNSUPPORTED;
  // algorithmic bytes: 3 source bytes gathered + 12 / 16 bytes written per output pixel
  fsd_prof::Scope prof(fsd_prof::kFirst, (double)a.total * (3.0 + (layout ? 16.0 : 12.0)), stream);
  FSD_LAUNCH(augment_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}
