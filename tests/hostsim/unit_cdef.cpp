// unit_cdef.cpp - TEST INFRASTRUCTURE ONLY.  The wavefront form of the CDEF strength search's distortion pass (tk_cdef.h: cdef_mse_block_wave - what the
// device runs: lanes = samples, then lanes = strengths) against the plain form (cdef_pass_mse: every (block, strength) filtered sample by sample with
// cdef_filter_px, which is pinned to the reference's cdef_filter_block by tests/golden/kat5.npz): the two mse[] arrays must be identical.  Random frames with
// flat / noisy / edge content, skipped blocks and all-skip filter blocks, sizes that leave partial filter blocks at the right and bottom edge, 8 / 10 / 12
// bits, the three search speeds.  Also the packed tap-offset table against cdef_dx / cdef_dy.  Host build of the engine headers (lanes as a loop).
#include <math.h>
#include "../../thor_amd/csrc/tk_cdef.h"
#include <random>
#include <vector>
namespace tk { Tables g_tab; }
using namespace tk;
static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } fails++; } } while (0)

template <typename PIX> static void run(int width, int height, int bitdepth, int speed, int damping, unsigned seed, int content) {
  std::mt19937 rng(seed);
  const int maxv = (1 << bitdepth) - 1;
  const int sy = width + 16, sc = width / 2 + 8;
  std::vector<PIX> src(sy * height + 2 * sc * (height / 2)), org(src.size()), rec(src.size());
  auto fill = [&](std::vector<PIX>& v, int noise) {
    for (int y = 0; y < height; y++)
      for (int x = 0; x < sy; x++) {
        int base = content == 0 ? (int)(rng() % (unsigned)(maxv + 1)) : content == 1 ? ((x / 5 + y / 3) & 1 ? maxv : 0) : (x * 3 + y * 2) * (maxv + 1) / 1024 % (maxv + 1);
        base += (int)(rng() % (unsigned)(2 * noise + 1)) - noise;
        v[y * sy + x] = (PIX)(base < 0 ? 0 : base > maxv ? maxv : base);
      }
    for (size_t k = (size_t)sy * height; k < v.size(); k++) {
      int base = content == 1 ? ((k / 7) & 1 ? maxv : 0) : (int)(rng() % (unsigned)(maxv + 1));
      v[k] = (PIX)base;
    }
  };
  fill(src, 3 << (bitdepth - 8));
  org = src;
  for (auto& p : org) { int v = (int)p + (int)(rng() % 9) - 4; p = (PIX)(v < 0 ? 0 : v > maxv ? maxv : v); }
  CdefJob<PIX> J;
  memset(&J, 0, sizeof(J));
  auto planes = [&](std::vector<PIX>& v) { Plane3<PIX> p; p.y = v.data(); p.u = v.data() + sy * height; p.v = p.u + sc * (height / 2); p.sy = sy; p.sc = sc; return p; };
  J.rec = planes(rec); J.src = planes(src); J.org = planes(org);
  J.width = width; J.height = height; J.bitdepth = bitdepth;
  J.cs = width / 4;
  std::vector<DbCell> cells((size_t)J.cs * (height / 4));
  memset(cells.data(), 0, cells.size() * sizeof(DbCell));
  J.nfb_h = (width + 63) / 64; J.nfb_v = (height + 63) / 64;
  const int nfb = J.nfb_h * J.nfb_v, nb = (width / 8) * (height / 8);
  for (int by = 0; by < height / 8; by++)
    for (int bx = 0; bx < width / 8; bx++) {
      const int fb = (by / 8) * J.nfb_h + bx / 8;
      const int m = (fb % 5 == 3) ? M_SKIP : ((rng() % 4) ? M_INTER : M_SKIP);   // some filter blocks entirely skipped
      for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) cells[(by * 2 + a) * J.cs + bx * 2 + b].mode = (uint8_t)m;
    }
  J.cells = cells.data();
  J.speed = speed; J.damping = damping; J.cdef_bits = 3; J.qp = 32;
  std::vector<int8_t> dir(nb);
  std::vector<int> var(nb), fbc(nfb), fbsel(nfb);
  std::vector<unsigned long long> mseA((size_t)2 * nfb * kCdefMaxStr), mseB(mseA.size());
  J.dir = dir.data(); J.var = var.data(); J.fb_compact = fbc.data(); J.fb_sel = fbsel.data();
  J.mse = mseA.data();
  cdef_pass_flags(J, 0, 1);
  cdef_pass_dir(J, 0, 1);
  if (content == 2) for (int b = 0; b < nb; b += 3) var[b] = 0;   // variance 0: adjusted primary strength 0 while the direction stays the block's
  cdef_pass_mse(J, 0, 1);
  J.mse = mseB.data();
  static CdefWaveWs<PIX> ws;
  const Team t = mk_team(0, 1);
  for (int b = 0; b < nb; b++) cdef_mse_block_wave(t, J, b, &ws);
  int ndiff = 0, nz = 0;
  for (size_t k = 0; k < mseA.size(); k++) { ndiff += mseA[k] != mseB[k]; nz += mseA[k] != 0; }
  CHECK(ndiff == 0, "%dx%d bd %d speed %d damping %d content %d: %d of %zu mse entries differ", width, height, bitdepth, speed, damping, content, ndiff, mseA.size());
  CHECK(nz > 0, "%dx%d bd %d speed %d: nothing was measured", width, height, bitdepth, speed);
}

int main() {
  for (int d = 0; d < 8; d++)
    for (int k = 0; k < 2; k++) CHECK(cdef_off12(d, k) == cdef_dy(d, k) * 12 + cdef_dx(d, k), "offset table d %d k %d", d, k);
  unsigned seed = 1;
  for (int content = 0; content < 3; content++)
    for (int speed = 0; speed < 3; speed++) {
      run<uint8_t>(200, 136, 8, speed, 6, seed++, content);    // partial filter blocks: 200 = 3 * 64 + 8, 136 = 2 * 64 + 8
      run<uint8_t>(128, 64, 8, speed, 5, seed++, content);
      run<uint16_t>(200, 136, 10, speed, 6, seed++, content);
      run<uint16_t>(144, 80, 12, speed, 6, seed++, content);
      run<uint16_t>(72, 72, 12, speed, 4, seed++, content);
    }
  if (fails) { fprintf(stderr, "%d failures\n", fails); return 1; }
  printf("ok\n");
  return 0;
}
