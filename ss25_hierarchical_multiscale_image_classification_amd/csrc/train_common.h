// Shared by train.hip (fp32) and train_amp.hip (fp16 mixed precision): the 20 convolutions of the ResNet18 encoder
// in the library's fixed order, and the layout of the flat parameter / gradient / running-statistics buffers.
#pragma once
#include "common.h"

namespace hipac {

struct ConvDesc {
  int cout, cin, ks, stride, hin, hout;
};
// 0 stem | per stage: b0.conv1, b0.conv2, [b0.downsample], b1.conv1, b1.conv2
static const ConvDesc kConvs[20] = {
    {64, 3, 7, 2, 224, 112},                                                                          // 0
    {64, 64, 3, 1, 56, 56},   {64, 64, 3, 1, 56, 56},   {64, 64, 3, 1, 56, 56},   {64, 64, 3, 1, 56, 56},    // 1-4
    {128, 64, 3, 2, 56, 28},  {128, 128, 3, 1, 28, 28}, {128, 64, 1, 2, 56, 28},                              // 5-7
    {128, 128, 3, 1, 28, 28}, {128, 128, 3, 1, 28, 28},                                                       // 8-9
    {256, 128, 3, 2, 28, 14}, {256, 256, 3, 1, 14, 14}, {256, 128, 1, 2, 28, 14},                             // 10-12
    {256, 256, 3, 1, 14, 14}, {256, 256, 3, 1, 14, 14},                                                       // 13-14
    {512, 256, 3, 2, 14, 7},  {512, 512, 3, 1, 7, 7},   {512, 256, 1, 2, 14, 7},                              // 15-17
    {512, 512, 3, 1, 7, 7},   {512, 512, 3, 1, 7, 7},                                                         // 18-19
};
constexpr int kNumConvs = 20;

static size_t conv_w_floats(int i) { return (size_t)kConvs[i].cout * kConvs[i].cin * kConvs[i].ks * kConvs[i].ks; }
static size_t param_offset(int i) {  // floats before conv i in the flat parameter buffer
  size_t o = 0;
  for (int k = 0; k < i; ++k) o += conv_w_floats(k) + 2 * (size_t)kConvs[k].cout;
  return o;
}
static size_t stat_offset(int i) {
  size_t o = 0;
  for (int k = 0; k < i; ++k) o += 2 * (size_t)kConvs[k].cout;
  return o;
}
static size_t packed_w_floats(int i) {  // [Cout][K] of the forward kernel (stem: 7 x 32 per row)
  return i == 0 ? (size_t)64 * 224 : conv_w_floats(i);
}


static inline unsigned grid_for(long long n, int cap = 4096) {
  long long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace hipac
