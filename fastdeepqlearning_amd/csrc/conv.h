// Implicit-GEMM convolutions of the pixel encoder (conv.hip; BASELINE config 5).  The reference's conv branch is dead code
// (franQ/Agent/components/encoder.py:16-23); the layer stack, layouts and parity target (torch conv2d) are this build's.
//
// No column matrix exists anywhere: a persistent workgroup (one per CU) walks GROUPS of images, a group's input maps go
// global -> LDS once by LDS-DMA (double-buffered: the next group lands while this one is multiplied), the MFMA operands are
// read straight out of the resident image through per-row patch offsets (a table built once per launch) plus compile-time
// tap offsets, and the layer's weights - or, for the weight gradients, the output block - stay in registers for the
// workgroup's whole life.
//   forward      out[img, oy, ox, :] = LeakyReLU(bias + W patch(img, oy, ox))                        NHWC, [nimg * OH * OW, Cout]
//   data grad    dprev[img, y, x, :] = LeakyReLU'(act_prev) * sum over the taps covering (y, x)       a gather: no col2im
//   weight grad  dW[co, k] = sum over (img, oy, ox) of dpre[.., co] * patch[.., k], db = column sums  partial slabs + reduction
// The FIRST layer reads uint8 NCHW frames as the ring stores them - a [T, B, c, h, w] uint8 batch or the ring's own block through
// a slot index per image (fdql_ring_window_slots) - and widens them in registers (K ordered (c, ky, kx));
// later layers read the previous layer's NHWC float32 output (K ordered (ky, kx, c)).  The weights [Cout, K] use the same order.
#pragma once
#include "common.h"
#include "update_kernels.h"

namespace fdql {

struct ConvSrc {
  const void *base;          // float32 NHWC maps [nimg][H][W][C], or uint8 NCHW frames (u8 = 1)
  int u8;
  const int *slots;          // u8 only.  null: frames [nimg][C*H*W] back to back.  Else `base` is a ring block [slots][C*H*W] and
};                           //   image i lives in slot slots[i] (fdql_ring_window_slots: (start[b] + t) % len for i = t B + b)

struct ConvFwdArgs {
  ConvSrc in;
  const float *W, *bias;     // [Cout][K], [Cout]
  float *out;                // [nimg][OH*OW][Cout], LeakyReLU applied
  long long nimg;
  ConvGeom g;
  int cout;
};

struct ConvDgradArgs {       // data gradient of a float32-input layer (the first layer has none)
  const float *dpre;         // [nimg][OH*OW][Cout] gradient of this layer's pre-activation
  const float *W;            // [Cout][K], K = (ky, kx, c)
  const float *act_prev;     // [nimg][H*W][C] the previous layer's output (gate reference)
  float *dprev;              // [nimg][H*W][C] gradient of the previous layer's pre-activation
  long long nimg;
  ConvGeom g;
  int cout;
};

struct ConvWgradArgs {
  ConvSrc in;
  const float *dpre;         // [nimg][OH*OW][Cout]
  float *wpart;              // [conv_wgrad_slabs()][Cout*K + Cout]: per-slab partial of (dW, db); summed by reduce_partials
  long long nimg;
  ConvGeom g;
  int cout;
};

// Which layers have a kernel instantiation (geometry is compile-time: tap offsets are instruction immediates).
bool conv_fwd_takes(const ConvGeom &g, int cout, bool u8);
bool conv_dgrad_takes(const ConvGeom &g, int cout);
bool conv_wgrad_takes(const ConvGeom &g, int cout, bool u8);
int conv_wgrad_slabs(const ConvGeom &g, int cout, bool u8, long long nimg);   // partial slabs the launch writes (0: not taken)
hipError_t conv_fwd_launch(const ConvFwdArgs &a, hipStream_t s);
hipError_t conv_dgrad_launch(const ConvDgradArgs &a, hipStream_t s);
hipError_t conv_wgrad_launch(const ConvWgradArgs &a, hipStream_t s);

}  // namespace fdql
