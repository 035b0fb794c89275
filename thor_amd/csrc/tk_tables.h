// tk_tables.h - host-side construction of the constant tables (tk::Tables).
// The zigzag scans and the HEVC-style integer DCT bases are generated from their
// defining rules rather than stored; tests/test_tables.py checks every entry
// against the reference's tables (common/common_tables.c:29-75, common/transform.c:37-241,
// enc/encode_tables.c:29-36, common/common_frame.c:37-45).
#pragma once
#include "tk_common.h"

namespace tk {

// squared_lambda_QP (enc/encode_tables.c:29-36): lambda^2 per QP, 4-decimal literals.
static const double kSquaredLambdaQP[52] = {
    0.0382,    0.0485,    0.0615,    0.0781,    0.0990,    0.1257,    0.1595,    0.2023,    0.2567,
    0.3257,    0.4132,    0.5243,    0.6652,    0.8440,    1.0709,    1.3588,    1.7240,    2.1874,
    2.7754,    3.5214,    4.4679,    5.6688,    7.1926,    9.1259,    11.5789,   14.6912,   18.6402,
    23.6505,   30.0076,   38.0735,   48.3075,   61.2922,   77.7672,   98.6706,   125.1926,  158.8437,
    201.5399,  255.7126,  324.4467,  411.6560,  522.3067,  662.6996,  840.8294,  1066.8393, 1353.5994,
    1717.4389, 2179.0763, 2764.7991, 3507.9607, 4450.8797, 5647.2498, 7165.1970};

static inline void make_zigzag(int n, int16_t* zz, int16_t* izz) {
  // Anti-diagonals d = i + j; odd diagonals run top-right -> bottom-left, even ones back.
  int idx = 0;
  for (int d = 0; d <= 2 * (n - 1); d++) {
    if (d & 1) {
      for (int i = 0; i < n; i++) {
        int j = d - i;
        if (j < 0 || j >= n) continue;
        zz[i * n + j] = (int16_t)idx;
        izz[idx++] = (int16_t)(i * n + j);
      }
    } else {
      for (int i = n - 1; i >= 0; i--) {
        int j = d - i;
        if (j < 0 || j >= n) continue;
        zz[i * n + j] = (int16_t)idx;
        izz[idx++] = (int16_t)(i * n + j);
      }
    }
  }
}

// HEVC core transform: entry (i, j) of the N-point basis is c[(i*(2j+1)*(32/N)) mod 128]
// where c[k] is the integer approximation of 64*sqrt(2)*cos(k*pi/64) (c[0] = 64).
static inline int hevc_cos(int k) {
  static const int mag[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                              61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
  k &= 127;
  if (k <= 32) return mag[k];
  if (k <= 64) return -mag[64 - k];
  if (k <= 96) return -mag[k - 64];
  return mag[128 - k];
}
static inline void make_dct(int n, int16_t* m) {
  int step = 32 / n;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) m[i * n + j] = (int16_t)hevc_cos(i * (2 * j + 1) * step);
}

static inline void init_tables(Tables* t) {
  make_zigzag(4, t->zz4, t->izz4);
  make_zigzag(8, t->zz8, t->izz8);
  make_zigzag(16, t->zz16, t->izz16);
  make_dct(4, t->dct4);
  make_dct(8, t->dct8);
  make_dct(16, t->dct16);
  make_dct(32, t->dct32);
  // chroma QP mapping (common_tables.c:68-73): identity to 29, then a compressed tail.
  static const uint8_t cq_tail[22] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36,
                                      36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45};
  {  // 8x8 inverse-quantiser step per qp used by the encoder_speed > 0 top-down split test (encode_block.c:2394)
    static const uint16_t iq[52] = {6,   7,   8,   8,   10,  11,  12,  13,  15,  17,  19,  21,  24,  27,  30,  34,  38,  43,
                                    48,  54,  60,  68,  76,  86,  96,  108, 121, 136, 152, 171, 192, 216, 242, 272, 305, 342,
                                    384, 431, 484, 543, 610, 684, 768, 862, 968, 1086, 1219, 1368, 1536, 1724, 1935, 2172};
    for (int q = 0; q < 52; q++) t->iq_8x8[q] = iq[q];
  }
  for (int q = 0; q < 30; q++) t->chroma_qp[q] = (uint8_t)q;
  for (int q = 30; q < 52; q++) t->chroma_qp[q] = cq_tail[q - 30];
  // deblocking thresholds (common_frame.c:37-45)
  for (int q = 0; q < 52; q++) t->beta[q] = (uint8_t)(q < 16 ? 0 : (q <= 28 ? q - 10 : 2 * q - 38));
  static const uint8_t tcv[56] = {0,  0,  1,  1,  2,  3,  4,  5,   6,   7,   8,   9,   10,  11,  12,  13,  14,  15, 16,
                                  17, 18, 20, 22, 24, 26, 28, 30,  32,  36,  40,  44,  48,  52,  56,  60,  64,  68, 72,
                                  80, 88, 96, 104, 112, 128, 144, 152, 160, 168, 176, 184, 192, 200, 208, 216, 224, 232};
  for (int q = 0; q < 56; q++) t->tc[q] = tcv[q];
}

}  // namespace tk
