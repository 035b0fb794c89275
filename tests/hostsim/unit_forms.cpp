// unit_forms.cpp - TEST INFRASTRUCTURE ONLY.  Algebra of the packed / vectorised forms the engine uses, checked against the plain
// per-sample definitions (which are the reference's formulas, pinned to its KATs elsewhere): the sub-pel samples as dot products
// (8-bit: subk8_sample / subk8_strip, 16-bit: subk16_sample / subk16_strip vs luma_sample), SSD as sum a^2 + sum b^2 - 2 sum ab modulo
// 2^32, the truncating average per dword, the saturating box sums row by row.  Host build of the engine headers (1-lane teams).
#include "../../thor_amd/csrc/tk_block.h"
#include <random>
namespace tk { Tables g_tab; long long g_prune_stat[8]; }
using namespace tk;
static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } fails++; } } while (0)

template <typename PIX> static void test_subpel(int bitdepth, unsigned seed) {
  std::mt19937 rng(seed);
  const int W = 64, H = 64, maxv = (1 << bitdepth) - 1;
  std::vector<PIX> plane(W * H);
  for (int mode = 0; mode < 3; mode++) {   // random, all-max, checkerboard of extremes
    for (int k = 0; k < W * H; k++) plane[k] = (PIX)(mode == 0 ? (int)(rng() % (unsigned)(maxv + 1)) : mode == 1 ? maxv : (((k % W) + (k / W)) & 1 ? maxv : 0));
    for (int bip = 0; bip < 2; bip++)
      for (int fy = 0; fy < 4; fy++)
        for (int fx = 0; fx < 4; fx++)
          for (int dy = 0; dy < 3; dy++)
            for (int dx = 0; dx < 3; dx++) {
              // a candidate whose integer position is (dy - 1, dx - 1) relative to the centre's, fractions (fy, fx)
              SubPel s;
              s.ver_int = dy - 1; s.hor_int = dx - 1; s.ver_frac = fy; s.hor_frac = fx;
              for (int m = 0; m < 6; m++) { s.th[m] = luma_tap(bip, fx, m); s.tv[m] = luma_tap(bip, fy, m); }
              s.ph = luma_taps8(bip, fx); s.pv = luma_taps8(bip, fy);   // the packed form of the same taps (what subk8_make reads since round 5)
              const int ci = 20, cj = 24;   // centre position (integer displacement 0) of the sample under test
              const PIX* ref = plane.data() + ci * W + cj;
              const int want = luma_sample<PIX>(ref, W, 0, 0, s, bip, bitdepth);
              if constexpr (sizeof(PIX) == 1) {
                unsigned long long wb[15];
                for (int q = 0; q < 15; q++) { unsigned long long v; memcpy(&v, ref + (q - 3) * W - 3, 8); wb[q] = v ^ 0x8080808080808080ull; }
                const SubK8 k = subk8_make(s, dy, dx, bip);
                CHECK(subk8_sample(wb, k) == want, "subk8_sample bip %d f %d,%d d %d,%d mode %d: %d != %d", bip, fy, fx, dy, dx, mode, subk8_sample(wb, k), want);
                int o8[8], sad = 0, sad_want = 0;
                for (int p = 0; p < 8; p++) { o8[p] = (int)(rng() & 255); sad_want += iabs(o8[p] - luma_sample<PIX>(ref, W, p, 0, s, bip, bitdepth)); }
                sad = subk8_strip(wb, k, o8, 0);
                CHECK(sad == sad_want, "subk8_strip bip %d f %d,%d d %d,%d mode %d: %d != %d", bip, fy, fx, dy, dx, mode, sad, sad_want);
              } else {
                uint32_t wb[15][4];
                for (int q = 0; q < 15; q++) memcpy(wb[q], ref + (q - 3) * W - 3, 16);
                const SubK16 k = subk16_make(s, dy, dx, bip);
                const int got = subk16_sample(wb, k, bitdepth);
                CHECK(got == want, "subk16_sample bd %d bip %d f %d,%d d %d,%d mode %d: %d != %d", bitdepth, bip, fy, fx, dy, dx, mode, got, want);
                int o8[8], sad_want = 0;
                for (int p = 0; p < 8; p++) { o8[p] = (int)(rng() % (unsigned)(maxv + 1)); sad_want += iabs(o8[p] - luma_sample<PIX>(ref, W, p, 0, s, bip, bitdepth)); }
                const int sad = subk16_strip(wb, k, o8, 0, bitdepth);
                CHECK(sad == sad_want, "subk16_strip bd %d bip %d f %d,%d d %d,%d mode %d: %d != %d", bitdepth, bip, fy, fx, dy, dx, mode, sad, sad_want);
              }
            }
  }
}

template <typename PIX> static void test_blocks(int bitdepth, unsigned seed) {
  std::mt19937 rng(seed);
  const Team t = mk_team(0, 1);
  const int maxv = (1 << bitdepth) - 1;
  for (int size : {4, 8, 16, 32, 64, 128})
    for (int mode = 0; mode < 3; mode++) {
      alignas(16) static PIX a[128 * 128], b[128 * 128], d[128 * 128 * 2], d2[128 * 128];
      for (int k = 0; k < size * size; k++) {
        a[k] = (PIX)(mode == 0 ? (int)(rng() % (unsigned)(maxv + 1)) : mode == 1 ? maxv : 0);
        b[k] = (PIX)(mode == 0 ? (int)(rng() % (unsigned)(maxv + 1)) : mode == 1 ? 0 : maxv);
      }
      // SSD: modular dot-product form vs the definition
      unsigned long long want = 0;
      for (int k = 0; k < size * size; k++) { long long e = (long long)a[k] - (long long)b[k]; want += (unsigned long long)(e * e); }
      const unsigned long long got = ssd_total(t, ssd_part<SP_GLOBAL, PIX>(t, a, size, b, size, size, size));
      // The modular form is exact while ONE LANE's share stays below 2^32: on the device a lane holds at most 128*128/64 = 256 samples
      // (256 x 4095^2 < 2^32, always exact).  A team smaller than a wavefront (this 1-lane team sums the whole block in one lane) takes the
      // 64-bit sample loop for blocks of 16-bit samples whose share can exceed that (round 5, tk_block.h:ssd_part): exact for every case.
      CHECK(got == want, "ssd size %d bd %d mode %d: %llu != %llu", size, bitdepth, mode, got, want);
      // truncating average
      average_yuv<SP_GLOBAL, PIX>(t, d, d + size * size, d + size * size * 5 / 4, a, a, a, b, b, b, size, size, size);   // chroma views overlap on purpose: only luma is checked
      int bad = 0;
      for (int k = 0; k < size * size; k++) bad += d[k] != (PIX)(((int)a[k] + (int)b[k]) >> 1);
      CHECK(bad == 0, "average size %d bd %d mode %d: %d samples differ", size, bitdepth, mode, bad);
      // block copy in pieces
      copy_block<SP_GLOBAL, SP_GLOBAL, PIX>(t, d2, size, a, size, size, size);
      CHECK(memcmp(d2, a, (size_t)size * size * sizeof(PIX)) == 0, "copy size %d", size);
      // 2*org - pred
      build_org8<PIX, SP_GLOBAL>(t, d2, a, size, b, size, bitdepth);
      bad = 0;
      for (int k = 0; k < size * size; k++) bad += d2[k] != (PIX)sat_pix(2 * (int)a[k] - (int)b[k], bitdepth);
      CHECK(bad == 0, "org8 size %d bd %d mode %d: %d samples differ", size, bitdepth, mode, bad);
    }
}

int main() {
  init_tables(&g_tab);
  test_subpel<uint8_t>(8, 1);
  test_subpel<uint16_t>(10, 2);
  test_subpel<uint16_t>(12, 3);
  test_blocks<uint8_t>(8, 4);
  test_blocks<uint16_t>(10, 5);
  test_blocks<uint16_t>(12, 6);
  if (fails) { fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
  printf("ok\n");
  return 0;
}
