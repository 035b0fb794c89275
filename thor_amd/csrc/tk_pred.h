// tk_pred.h - inter (sub-pel) and intra prediction, team-cooperative.
// Scalar specifications followed: common/inter_prediction.c:51-63 (clip_mv), :65-115
// (chroma 1/8-pel 4-tap), :117-181 (luma 1/4-pel 6-tap + centre filter), :185-226
// (get_inter_prediction_yuv), :228-247 (average_blocks_all); common/intra_prediction.c:57-183
// (make_top_and_left) and :185-428 (the ten predictors); filter taps common_kernels.c:1905-1928.
#pragma once
#include "tk_common.h"

namespace tk {

// Filter taps as IMMEDIATES: a table indexed at run time is a constant-memory array - every luma_setup of the sub-pel search and of the
// trial predictions paid a global-memory round trip for its twelve taps (round 5: 4 global_load + 12 readfirstlane per set-up).  Packed into one
// 64-bit literal per fraction (tap m = int8 lane m) the look-up is three selects on the fraction.
TK_DEV constexpr unsigned long long pack_taps6(int a, int b, int c, int d, int e, int f) {
  return (unsigned long long)(unsigned)(a & 0xff) | ((unsigned long long)(unsigned)(b & 0xff) << 8) | ((unsigned long long)(unsigned)(c & 0xff) << 16) |
         ((unsigned long long)(unsigned)(d & 0xff) << 24) | ((unsigned long long)(unsigned)(e & 0xff) << 32) | ((unsigned long long)(unsigned)(f & 0xff) << 40);
}
// luma taps, [bipred set][frac][6]  (common_kernels.c:1905-1917), tap m in bits [8m, 8m + 8) as int8
TK_DEV unsigned long long luma_taps8(int bipred, int frac) {
  // standard: {0,0,64,0,0,0},{1,-7,55,19,-5,1},{1,-7,38,38,-7,1},{1,-5,19,55,-7,1}
  // bipred  : {0,0,64,0,0,0},{2,-10,59,17,-5,1},{1,-8,39,39,-8,1},{1,-5,17,59,-10,2}
  const unsigned long long z = pack_taps6(0, 0, 64, 0, 0, 0);
  const unsigned long long s1 = pack_taps6(1, -7, 55, 19, -5, 1), s2 = pack_taps6(1, -7, 38, 38, -7, 1), s3 = pack_taps6(1, -5, 19, 55, -7, 1);
  const unsigned long long b1 = pack_taps6(2, -10, 59, 17, -5, 1), b2 = pack_taps6(1, -8, 39, 39, -8, 1), b3 = pack_taps6(1, -5, 17, 59, -10, 2);
  const unsigned long long s = frac == 1 ? s1 : frac == 2 ? s2 : s3, b = frac == 1 ? b1 : frac == 2 ? b2 : b3;
  return frac == 0 ? z : (bipred ? b : s);
}
TK_DEV int luma_tap(int bipred, int frac, int m) { return (int)(int8_t)(luma_taps8(bipred, frac) >> (8 * m)); }
// chroma taps [frac][4] (common_kernels.c:1919-1928), tap m in bits [8m, 8m + 8) as int8
TK_DEV unsigned chroma_taps8(int frac) {
  auto pk = [](int a, int b, int c, int d) -> unsigned { return (unsigned)(a & 0xff) | ((unsigned)(b & 0xff) << 8) | ((unsigned)(c & 0xff) << 16) | ((unsigned)(d & 0xff) << 24); };
  const unsigned lo = frac == 0 ? pk(0, 64, 0, 0) : frac == 1 ? pk(-2, 58, 10, -2) : frac == 2 ? pk(-4, 54, 16, -2) : pk(-4, 44, 28, -4);
  const unsigned hi = frac == 4 ? pk(-4, 36, 36, -4) : frac == 5 ? pk(-4, 28, 44, -4) : frac == 6 ? pk(-2, 16, 54, -4) : pk(-2, 10, 58, -2);
  return frac < 4 ? lo : hi;
}
TK_DEV int chroma_tap(int frac, int m) { return (int)(int8_t)(chroma_taps8(frac) >> (8 * m)); }

// clip_mv (inter_prediction.c:51-63)
TK_DEV mv_t clip_mv(mv_t mv, int ypos, int xpos, int fwidth, int fheight, int bwidth, int bheight, int sign) {
  const int ext = kPadY - 16;
  int mvy = sign ? -mv.y : mv.y;
  int mvx = sign ? -mv.x : mv.x;
  if (ypos + mvy / 4 < -ext) mvy = 4 * (-ext - ypos);
  if (ypos + mvy / 4 + bheight > fheight + ext) mvy = 4 * (fheight + ext - ypos - bheight);
  if (xpos + mvx / 4 < -ext) mvx = 4 * (-ext - xpos);
  if (xpos + mvx / 4 + bwidth > fwidth + ext) mvx = 4 * (fwidth + ext - xpos - bwidth);
  return mk_mv(sign ? -mvx : mvx, sign ? -mvy : mvy);
}

// Sub-pel set-up of one PU / candidate: integer displacement, fractions and the two 6-tap filters
// (hoisted out of the per-sample code: it is uniform per candidate).
struct SubPel {
  int hor_int, ver_int, hor_frac, ver_frac;
  int th[6], tv[6];
  unsigned long long ph, pv;   // the same taps packed (int8 lane m = tap m): what the 8-bit dot-product forms consume
};
TK_DEV SubPel luma_setup(mv_t mv, int sign, int width, int height, int pic_w, int pic_h, int xpos, int ypos, int bipred = 0) {
  int mx = sign ? -mv.x : mv.x, my = sign ? -mv.y : mv.y;
  SubPel s;
  s.ver_frac = my & 3;
  s.hor_frac = mx & 3;
  int vi = my >> 2, hi = mx >> 2;
  vi = tmin(vi, pic_h - ypos);
  vi = tmax(vi, -xpos - height);  // sic: xpos (inter_prediction.c:129), kept for parity
  hi = tmin(hi, pic_w - xpos);
  hi = tmax(hi, -xpos - width);
  s.ver_int = vi;
  s.hor_int = hi;
  const unsigned long long ph = luma_taps8(bipred, s.hor_frac), pv = luma_taps8(bipred, s.ver_frac);
  s.ph = ph; s.pv = pv;
  for (int m = 0; m < 6; m++) {
    s.th[m] = (int)(int8_t)(ph >> (8 * m));
    s.tv[m] = (int)(int8_t)(pv >> (8 * m));
  }
  return s;
}

// Six consecutive samples starting at p (any alignment) with wide loads.  Reads up to 2 samples past the
// sixth: always inside the padded reference planes (160-sample borders + 64 samples of slack at the end).
template <typename PIX> TK_DEV void load6(const PIX* p, int r[6]) {
  if (sizeof(PIX) == 1) {
    const unsigned long long v = gload64(p);
    for (int m = 0; m < 6; m++) r[m] = (int)((v >> (8 * m)) & 0xffu);
  } else {
    const unsigned long long v0 = gload64(p);
    const unsigned v1 = gload32((const char*)p + 8);
    for (int m = 0; m < 4; m++) r[m] = (int)((v0 >> (16 * m)) & 0xffffu);
    r[4] = (int)(v1 & 0xffffu);
    r[5] = (int)(v1 >> 16);
  }
}

template <typename PIX> TK_DEV void load4(const PIX* p, int r[4]) {
  if (sizeof(PIX) == 1) {
    const unsigned v = gload32(p);
    for (int m = 0; m < 4; m++) r[m] = (int)((v >> (8 * m)) & 0xffu);
  } else {
    const unsigned long long v = gload64(p);
    for (int m = 0; m < 4; m++) r[m] = (int)((v >> (16 * m)) & 0xffffu);
  }
}

// One luma prediction sample at (i=row, j=col) of a PU whose reference pointer (at integer
// displacement 0) is `ref`.
template <typename PIX>
TK_DEV int luma_sample(const PIX* ref, int stride, int i, int j, const SubPel& s, int bipred, int bitdepth) {
  const TK_GLOBAL PIX* p = gptr(ref) + (i + s.ver_int) * stride + (j + s.hor_int);
  if (s.ver_frac == 0 && s.hor_frac == 0) return p[0];
  if (s.ver_frac == 2 && s.hor_frac == 2 && bipred < 2) {
    int a[4], b[4], c[4], d[4];  // rows -1 .. 2, columns -1 .. 2
    load4((const PIX*)(p - stride - 1), a); load4((const PIX*)(p - 1), b); load4((const PIX*)(p + stride - 1), c); load4((const PIX*)(p + 2 * stride - 1), d);
    int sum = a[1] + a[2] + b[0] + 2 * b[1] + 2 * b[2] + b[3] + c[0] + 2 * c[1] + 2 * c[2] + c[3] + d[1] + d[2];
    return sat_pix((sum + 8) >> 4, bitdepth);
  }
  if (s.hor_frac == 0) {
    int sum = 0;
    for (int m = 0; m < 6; m++) sum += s.tv[m] * p[(m - 2) * stride];
    return sat_pix((sum * 64 + 2048) >> 12, bitdepth);
  }
  if (s.ver_frac == 0) {
    int r[6];
    load6((const PIX*)(p - 2), r);
    int sum = 0;
    for (int m = 0; m < 6; m++) sum += s.th[m] * r[m];
    return sat_pix((sum * 64 + 2048) >> 12, bitdepth);
  }
  // 2-D: six rows of six samples; each row is fetched with one (u8) or two (u16) wide loads instead of six
  // narrow ones, and all rows are in flight before the first multiply
  int r[6][6];
  for (int m = 0; m < 6; m++) load6((const PIX*)(p + (m - 2) * stride - 2), r[m]);
  int sum = 0;
  for (int n = 0; n < 6; n++) {
    int col = 0;
    for (int m = 0; m < 6; m++) col += s.tv[m] * r[m][n];
    sum += s.th[n] * col;
  }
  return sat_pix((sum + 2048) >> 12, bitdepth);
}

// ---------------------------------------------------------------------------------
// Register-resident 8x8 reference window for the sub-pel search.  The eight half- (quarter-) pel candidates
// around one centre all interpolate from integer positions within +-1 sample of the centre's, i.e. from the
// same 8 rows x 8 columns of the reference (taps -2..+3 around -1..+1).  A lane therefore fetches that window
// ONCE per sample (8 wide loads in flight together) and evaluates all eight candidates from registers instead
// of issuing six dependent row loads per candidate and sample.
// ---------------------------------------------------------------------------------
template <typename PIX> struct WinRow;
template <> struct WinRow<uint8_t> { unsigned long long a; };
template <> struct WinRow<uint16_t> { unsigned long long a, b; };
TK_DEV void win_load(const uint8_t* p, WinRow<uint8_t>& r) { r.a = gload64(p); }
TK_DEV void win_load(const uint16_t* p, WinRow<uint16_t>& r) { r.a = gload64(p); r.b = gload64(p + 4); }
// row selected by dy in {0,1,2} (uniform) out of three consecutive rows, then shifted right by dx samples
TK_DEV WinRow<uint8_t> win_pick(const WinRow<uint8_t>& r0, const WinRow<uint8_t>& r1, const WinRow<uint8_t>& r2, int dy, int dx) {
  WinRow<uint8_t> o;
  o.a = (dy == 0 ? r0.a : (dy == 1 ? r1.a : r2.a)) >> (8 * dx);
  return o;
}
TK_DEV WinRow<uint16_t> win_pick(const WinRow<uint16_t>& r0, const WinRow<uint16_t>& r1, const WinRow<uint16_t>& r2, int dy, int dx) {
  const unsigned long long a = dy == 0 ? r0.a : (dy == 1 ? r1.a : r2.a), b = dy == 0 ? r0.b : (dy == 1 ? r1.b : r2.b);
  WinRow<uint16_t> o;
  if (dx == 0) { o.a = a; o.b = b; }
  else { o.a = (a >> (16 * dx)) | (b << (64 - 16 * dx)); o.b = b >> (16 * dx); }
  return o;
}
TK_DEV int win_at(const WinRow<uint8_t>& r, int n) { return (int)((r.a >> (8 * n)) & 0xffu); }          // n = 0..5
TK_DEV int win_at(const WinRow<uint16_t>& r, int n) { return n < 4 ? (int)((r.a >> (16 * n)) & 0xffffu) : (int)((r.b >> (16 * (n - 4))) & 0xffffu); }

// luma_sample() evaluated on six picked rows (row m, column n <-> p[(m-2)*stride + (n-2)])
template <typename PIX> TK_DEV int luma_sample_win(const WinRow<PIX> w[6], const SubPel& s, int bipred, int bitdepth) {
  if (s.ver_frac == 0 && s.hor_frac == 0) return win_at(w[2], 2);
  if (s.ver_frac == 2 && s.hor_frac == 2 && bipred < 2) {
    int sum = win_at(w[1], 2) + win_at(w[1], 3) + win_at(w[2], 1) + 2 * win_at(w[2], 2) + 2 * win_at(w[2], 3) + win_at(w[2], 4) +
              win_at(w[3], 1) + 2 * win_at(w[3], 2) + 2 * win_at(w[3], 3) + win_at(w[3], 4) + win_at(w[4], 2) + win_at(w[4], 3);
    return sat_pix((sum + 8) >> 4, bitdepth);
  }
  if (s.hor_frac == 0) {
    int sum = 0;
    for (int m = 0; m < 6; m++) sum += s.tv[m] * win_at(w[m], 2);
    return sat_pix((sum * 64 + 2048) >> 12, bitdepth);
  }
  if (s.ver_frac == 0) {
    int sum = 0;
    for (int n = 0; n < 6; n++) sum += s.th[n] * win_at(w[2], n);
    return sat_pix((sum * 64 + 2048) >> 12, bitdepth);
  }
  int sum = 0;
  for (int n = 0; n < 6; n++) {
    int col = 0;
    for (int m = 0; m < 6; m++) col += s.tv[m] * win_at(w[m], n);
    sum += s.th[n] * col;
  }
  return sat_pix((sum + 2048) >> 12, bitdepth);
}

// 8-bit samples: the same evaluation with packed dot products.  The window rows are kept XOR 0x80 per byte
// ("sample - 128" as int8), the horizontal taps packed as four + two int8; a six-tap row sum is then two
// v_dot4_i32_i8 and the missing 128 * (sum of taps) = 128 * 64 is added back - the same integer as the 6 MACs.
TK_DEV int dot4_i8(int a, int b, int c) {
#if TK_HOST
  for (int k = 0; k < 4; k++) c += (int)(int8_t)(a >> (8 * k)) * (int)(int8_t)(b >> (8 * k));
  return c;
#else
  return __builtin_amdgcn_sdot4(a, b, c, false);
#endif
}
// Sub-pel search, 8-bit: one prediction sample straight from the eight biased window rows (row q = reference row
// centre.ver_int - 3 + q, byte n = column centre.hor_int - 3 + n) for a candidate at integer offset (dy, dx) in 0..2 from
// centre - 1.  The horizontal taps sit in an 8-byte vector at byte offset dx, so a row sum is two v_dot4 on the row as it was
// loaded (no 64-bit shifts); every fractional position except the (1/2, 1/2) centre filter is the general separable form
// (a zero fraction has taps {0,0,64,0,0,0}, which reproduces the reference's 1-D formulas and the full-pel copy exactly).
struct SubK8 {
  unsigned long long th8;  // horizontal taps (int8 lanes) at bytes dx .. dx+5
  int tv[6];
  int centre, dy, dx;
};
TK_DEV SubK8 subk8_make(const SubPel& s, int dy, int dx, int bipred) {
  SubK8 k;
  k.th8 = s.ph << (8 * dx);
  for (int m = 0; m < 6; m++) k.tv[m] = (int)(int8_t)(s.pv >> (8 * m));
  k.centre = s.ver_frac == 2 && s.hor_frac == 2 && bipred < 2;
  k.dy = dy; k.dx = dx;
  return k;
}
TK_DEV int mul24(int a, int b) {
#if TK_HOST
  return a * b;
#else
  return __mul24(a, b);
#endif
}
template <int DY> TK_DEV int subk8_sample_dy(const unsigned long long* wr, const SubK8& k) {
  if (k.centre) {  // 12-tap centre filter (inter_prediction.c:146-160) as four rows of byte weights
    const unsigned long long wA = 0x0000000001010000ull << (8 * k.dx), wB = 0x0000000102020100ull << (8 * k.dx);
    int sum = 128 * 16;
    sum = dot4_i8((int)(unsigned)wA, (int)(unsigned)wr[DY + 1], dot4_i8((int)(unsigned)(wA >> 32), (int)(unsigned)(wr[DY + 1] >> 32), sum));
    sum = dot4_i8((int)(unsigned)wB, (int)(unsigned)wr[DY + 2], dot4_i8((int)(unsigned)(wB >> 32), (int)(unsigned)(wr[DY + 2] >> 32), sum));
    sum = dot4_i8((int)(unsigned)wB, (int)(unsigned)wr[DY + 3], dot4_i8((int)(unsigned)(wB >> 32), (int)(unsigned)(wr[DY + 3] >> 32), sum));
    sum = dot4_i8((int)(unsigned)wA, (int)(unsigned)wr[DY + 4], dot4_i8((int)(unsigned)(wA >> 32), (int)(unsigned)(wr[DY + 4] >> 32), sum));
    return sat_pix((sum + 8) >> 4, 8);
  }
  int sum = 0;
#if !TK_HOST
#pragma unroll
#endif
  for (int m = 0; m < 6; m++) {
    const unsigned long long row = wr[DY + m];
    const int h = dot4_i8((int)(unsigned)k.th8, (int)(unsigned)row, dot4_i8((int)(unsigned)(k.th8 >> 32), (int)(unsigned)(row >> 32), 128 * 64));
    sum += mul24(k.tv[m], h);
  }
  return sat_pix((sum + 2048) >> 12, 8);
}
TK_DEV int subk8_sample(const unsigned long long* wr, const SubK8& k) {
  if (k.dy == 0) return subk8_sample_dy<0>(wr, k);
  if (k.dy == 1) return subk8_sample_dy<1>(wr, k);
  return subk8_sample_dy<2>(wr, k);
}

// The same for a vertical strip of eight samples of one column (PUs of 512 samples and more): the 13 row sums the eight
// samples need are formed once (26 v_dot4 instead of 96) and 15 row loads serve eight samples instead of eight loads each.
// wb: 15 biased rows starting at reference row centre.ver_int - 3 + (first sample row); o: the eight original samples.
template <int DY> TK_DEV int subk8_strip_dy(const unsigned long long* wb, const SubK8& k, const int* o, int sad) {
  auto ad = [](int x, int y) -> int { const unsigned a = (unsigned)x, b = (unsigned)y; return (int)((a > b ? a : b) - (a < b ? a : b)); };
  if (k.centre) {
    const unsigned long long wA = 0x0000000001010000ull << (8 * k.dx), wB = 0x0000000102020100ull << (8 * k.dx);
    int ra[12], rb[12];
#if !TK_HOST
#pragma unroll
#endif
    for (int q = 1; q < 12; q++) {
      const unsigned long long row = wb[DY + q];
      ra[q] = dot4_i8((int)(unsigned)wA, (int)(unsigned)row, dot4_i8((int)(unsigned)(wA >> 32), (int)(unsigned)(row >> 32), 0));
      rb[q] = dot4_i8((int)(unsigned)wB, (int)(unsigned)row, dot4_i8((int)(unsigned)(wB >> 32), (int)(unsigned)(row >> 32), 0));
    }
#if !TK_HOST
#pragma unroll
#endif
    for (int p = 0; p < 8; p++) sad += ad(o[p], sat_pix((128 * 16 + ra[p + 1] + rb[p + 2] + rb[p + 3] + ra[p + 4] + 8) >> 4, 8));
    return sad;
  }
  int hs[13];
#if !TK_HOST
#pragma unroll
#endif
  for (int q = 0; q < 13; q++) {
    const unsigned long long row = wb[DY + q];
    hs[q] = dot4_i8((int)(unsigned)k.th8, (int)(unsigned)row, dot4_i8((int)(unsigned)(k.th8 >> 32), (int)(unsigned)(row >> 32), 128 * 64));
  }
#if !TK_HOST
#pragma unroll
#endif
  for (int p = 0; p < 8; p++) {
    int sum = 0;
#if !TK_HOST
#pragma unroll
#endif
    for (int m = 0; m < 6; m++) sum += mul24(k.tv[m], hs[p + m]);
    sad += ad(o[p], sat_pix((sum + 2048) >> 12, 8));
  }
  return sad;
}
TK_DEV int subk8_strip(const unsigned long long* wb, const SubK8& k, const int* o, int sad) {
  if (k.dy == 0) return subk8_strip_dy<0>(wb, k, o, sad);
  if (k.dy == 1) return subk8_strip_dy<1>(wb, k, o, sad);
  return subk8_strip_dy<2>(wb, k, o, sad);
}

// 16-bit samples: the same sub-pel evaluation with packed int16 dot products (v_dot2_i32_i16).  A window row is eight samples
// in four dwords (row q = reference row centre.ver_int - 3 + q, sample n = column centre.hor_int - 3 + n); a candidate at integer
// offset (dy, dx) in 0..2 from centre - 1 takes samples dx .. dx+5 of rows dy .. dy+5 as three sample pairs - the dwords as
// loaded for an even dx, one v_alignbyte per pair for dx = 1 - against the horizontal taps packed in pairs: a six-tap row sum is
// three v_dot2, the vertical pass six v_mad_i32_i24 (|row sum| <= 94 * 4095 < 2^19).  Every fractional position except the
// (1/2, 1/2) centre filter is this one separable form (see SubK8).
struct SubK16 {
  uint32_t th2[3];   // horizontal taps (t0,t1) (t2,t3) (t4,t5), low half = first tap
  int tv[6];
  int centre, dy, dx;
};
TK_DEV SubK16 subk16_make(const SubPel& s, int dy, int dx, int bipred) {
  SubK16 k;
  for (int q = 0; q < 3; q++) k.th2[q] = (uint32_t)(uint16_t)(int16_t)s.th[2 * q] | ((uint32_t)(uint16_t)(int16_t)s.th[2 * q + 1] << 16);
  for (int m = 0; m < 6; m++) k.tv[m] = s.tv[m];
  k.centre = s.ver_frac == 2 && s.hor_frac == 2 && bipred < 2;
  k.dy = dy; k.dx = dx;
  return k;
}
TK_DEV int dot2_i16(uint32_t a, uint32_t b, int c) {
#if TK_HOST
  return c + (int)(int16_t)(a & 0xffffu) * (int)(int16_t)(b & 0xffffu) + (int)(int16_t)(a >> 16) * (int)(int16_t)(b >> 16);
#else
  typedef short __attribute__((ext_vector_type(2))) s16x2;
  s16x2 x, y;
  __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
  return __builtin_amdgcn_sdot2(x, y, c, false);
#endif
}
// sample pair q (0..2) of a row (four dwords d) for a candidate at horizontal offset DX
template <int DX> TK_DEV uint32_t row_pair16(const uint32_t* d, int q) {
  if constexpr (DX == 0) return d[q];
  else if constexpr (DX == 2) return d[q + 1];
  else {
#if TK_HOST
    return (d[q] >> 16) | (d[q + 1] << 16);
#else
    return __builtin_amdgcn_alignbyte(d[q + 1], d[q], 2);
#endif
  }
}
// rows: 8 x 4 dwords (row-major)
template <int DY, int DX> TK_DEV int subk16_sample_t(const uint32_t (*rows)[4], const SubK16& k, int bitdepth) {
  if (k.centre) {  // 12-tap centre filter (inter_prediction.c:146-160): rows 1 and 4 weigh columns {0,0,1,1,0,0}, rows 2 and 3 {0,1,2,2,1,0}
    int sum = 8;
    sum = dot2_i16(0x00010001u, row_pair16<DX>(rows[DY + 1], 1), sum);
    sum = dot2_i16(0x00010001u, row_pair16<DX>(rows[DY + 4], 1), sum);
#if !TK_HOST
#pragma unroll
#endif
    for (int m = 2; m <= 3; m++) {
      sum = dot2_i16(0x00010000u, row_pair16<DX>(rows[DY + m], 0), sum);
      sum = dot2_i16(0x00020002u, row_pair16<DX>(rows[DY + m], 1), sum);
      sum = dot2_i16(0x00000001u, row_pair16<DX>(rows[DY + m], 2), sum);
    }
    return sat_pix(sum >> 4, bitdepth);
  }
  int sum = 2048;
#if !TK_HOST
#pragma unroll
#endif
  for (int m = 0; m < 6; m++) {
    const uint32_t* d = rows[DY + m];
    const int h = dot2_i16(k.th2[0], row_pair16<DX>(d, 0), dot2_i16(k.th2[1], row_pair16<DX>(d, 1), dot2_i16(k.th2[2], row_pair16<DX>(d, 2), 0)));
    sum += mul24(k.tv[m], h);
  }
  return sat_pix(sum >> 12, bitdepth);
}
TK_DEV int subk16_sample(const uint32_t (*rows)[4], const SubK16& k, int bitdepth) {
  // (dy, dx) are wave-uniform: one scalar branch picks the straight-line instance
  switch (k.dy * 3 + k.dx) {
    case 0: return subk16_sample_t<0, 0>(rows, k, bitdepth);
    case 1: return subk16_sample_t<0, 1>(rows, k, bitdepth);
    case 2: return subk16_sample_t<0, 2>(rows, k, bitdepth);
    case 3: return subk16_sample_t<1, 0>(rows, k, bitdepth);
    case 4: return subk16_sample_t<1, 1>(rows, k, bitdepth);
    case 5: return subk16_sample_t<1, 2>(rows, k, bitdepth);
    case 6: return subk16_sample_t<2, 0>(rows, k, bitdepth);
    case 7: return subk16_sample_t<2, 1>(rows, k, bitdepth);
    default: return subk16_sample_t<2, 2>(rows, k, bitdepth);
  }
}

// The same for a vertical strip of eight samples of one column (16-bit PUs of 512 samples and more; see subk8_strip): 13 row sums
// (39 v_dot2) serve eight samples instead of 144, 15 row loads instead of 64.  wb: 15 rows of four dwords starting at reference row
// centre.ver_int - 3 + (first sample row); o: the eight original samples.
template <int DY, int DX> TK_DEV int subk16_strip_t(const uint32_t (*wb)[4], const SubK16& k, const int* o, int sad, int bitdepth) {
  if (k.centre) {
    int ra[12], rb[12];
#if !TK_HOST
#pragma unroll
#endif
    for (int q = 1; q < 12; q++) {
      const uint32_t* d = wb[DY + q];
      ra[q] = dot2_i16(0x00010001u, row_pair16<DX>(d, 1), 0);
      rb[q] = dot2_i16(0x00010000u, row_pair16<DX>(d, 0), dot2_i16(0x00020002u, row_pair16<DX>(d, 1), dot2_i16(0x00000001u, row_pair16<DX>(d, 2), 0)));
    }
#if !TK_HOST
#pragma unroll
#endif
    for (int p = 0; p < 8; p++) sad += iabs(o[p] - sat_pix((8 + ra[p + 1] + rb[p + 2] + rb[p + 3] + ra[p + 4]) >> 4, bitdepth));
    return sad;
  }
  int hs[13];
#if !TK_HOST
#pragma unroll
#endif
  for (int q = 0; q < 13; q++) {
    const uint32_t* d = wb[DY + q];
    hs[q] = dot2_i16(k.th2[0], row_pair16<DX>(d, 0), dot2_i16(k.th2[1], row_pair16<DX>(d, 1), dot2_i16(k.th2[2], row_pair16<DX>(d, 2), 0)));
  }
#if !TK_HOST
#pragma unroll
#endif
  for (int p = 0; p < 8; p++) {
    int sum = 2048;
#if !TK_HOST
#pragma unroll
#endif
    for (int m = 0; m < 6; m++) sum += mul24(k.tv[m], hs[p + m]);
    sad += iabs(o[p] - sat_pix(sum >> 12, bitdepth));
  }
  return sad;
}
TK_DEV int subk16_strip(const uint32_t (*wb)[4], const SubK16& k, const int* o, int sad, int bitdepth) {
  switch (k.dy * 3 + k.dx) {
    case 0: return subk16_strip_t<0, 0>(wb, k, o, sad, bitdepth);
    case 1: return subk16_strip_t<0, 1>(wb, k, o, sad, bitdepth);
    case 2: return subk16_strip_t<0, 2>(wb, k, o, sad, bitdepth);
    case 3: return subk16_strip_t<1, 0>(wb, k, o, sad, bitdepth);
    case 4: return subk16_strip_t<1, 1>(wb, k, o, sad, bitdepth);
    case 5: return subk16_strip_t<1, 2>(wb, k, o, sad, bitdepth);
    case 6: return subk16_strip_t<2, 0>(wb, k, o, sad, bitdepth);
    case 7: return subk16_strip_t<2, 1>(wb, k, o, sad, bitdepth);
    default: return subk16_strip_t<2, 2>(wb, k, o, sad, bitdepth);
  }
}

// get_inter_prediction_luma for a whole PU (team-parallel over samples).
template <int SP, typename PIX>
TK_DEV void pred_luma(const Team t, PIX* dst_, int dstride, const PIX* ref, int rstride, int width, int height, mv_t mv,
                      int sign, int bipred, int pic_w, int pic_h, int xpos, int ypos, int bitdepth) {
  const auto dst = spc<SP>(dst_);
  SubPel s = luma_setup(mv, sign, width, height, pic_w, pic_h, xpos, ypos, bipred);
  const Div2 pw = mk_div(width);  // rectangular frame-edge skip blocks have non power-of-two widths
  for (int k = t.rank; k < width * height; k += t.size) {
    int i, j;
    split2(pw, k, i, j);
    dst[i * dstride + j] = (PIX)luma_sample(ref, rstride, i, j, s, bipred, bitdepth);
  }
}

template <int SP, typename PIX>
TK_DEV void pred_chroma(const Team t, PIX* dst_, int dstride, const PIX* ref, int rstride, int width, int height,
                        mv_t mv, int sign, int pic_w2, int pic_h2, int xpos, int ypos, int bitdepth) {
  const auto dst = spc<SP>(dst_);
  int mx = sign ? -mv.x : mv.x, my = sign ? -mv.y : mv.y;
  int vf = my & 7, hf = mx & 7;
  int vi = my >> 3, hi = mx >> 3;
  vi = tmin(vi, pic_h2 - ypos);
  vi = tmax(vi, -xpos - height);  // sic (inter_prediction.c:78)
  hi = tmin(hi, pic_w2 - xpos);
  hi = tmax(hi, -xpos - width);
  // the eight taps once per PU (they are selects on the fractions, not table reads: keep them out of the sample loop)
  int th[4], tv[4];
  {
    const unsigned ph = chroma_taps8(hf), pv = chroma_taps8(vf);
    for (int n = 0; n < 4; n++) { th[n] = (int)(int8_t)(ph >> (8 * n)); tv[n] = (int)(int8_t)(pv >> (8 * n)); }
  }
  for (int k = t.rank; k < width * height; k += t.size) {
    int i, j;
    split2(mk_div(width), k, i, j);
    const TK_GLOBAL PIX* p = gptr(ref) + (i + vi) * rstride + (j + hi);
    int v;
    if (vf == 0 && hf == 0) {
      v = p[0];
    } else {
      int sum = 0;
      for (int n = 0; n < 4; n++) {
        const TK_GLOBAL PIX* q = p + (n - 1) * rstride;
        int row = th[0] * q[-1] + th[1] * q[0] + th[2] * q[1] + th[3] * q[2];
        sum += tv[n] * row;
      }
      v = sat_pix((sum + 2048) >> 12, bitdepth);
    }
    dst[i * dstride + j] = (PIX)v;
  }
}

// get_inter_prediction_yuv (inter_prediction.c:185-226), 4:2:0.  dst planes are compact blocks
// of stride `size` (luma) / size/2 (chroma).  `split`: 1 => four quadrant PUs with mv_arr[0..3].
// SP: address space of the destination blocks py / pu / pv
#if defined(TK_PIY_INLINE)   // A/B switch (round 6): inline the block prediction into its callers instead of a call that saves 21-27 callee-saved VGPRs
#define TK_PIY TK_DEV
#else
#define TK_PIY TK_DEVNI
#endif
template <int SP, typename PIX>
TK_PIY void pred_inter_yuv(const Team t, const Plane3<PIX> ref, PIX* py, PIX* pu, PIX* pv, int ypos, int xpos,
                           int size, int bw, int bh, const mv_t* mv_arr, int sign, int pic_w, int pic_h,
                           int enable_bipred, int split, int bitdepth, int luma_only = 0) {
  const int div = split + 1;
  const int bwidth = bw / div, bheight = bh / div;
  const int pstride = size;
  const int yc = ypos >> 1, xc = xpos >> 1;
  const PIX* ry = ref.y + ypos * ref.sy + xpos;
  const PIX* ru = ref.u + yc * ref.sc + xc;
  const PIX* rv = ref.v + yc * ref.sc + xc;
  for (int index = 0; index < div * div; index++) {
    int idx = index & 1, idy = (index >> 1) & 1;
    int offpY = idy * bheight * pstride + idx * bwidth;
    int offpC = ((idy * bheight * pstride) >> 2) + ((idx * bwidth) >> 1);
    int offrY = idy * bheight * ref.sy + idx * bwidth;
    int offrC = ((idy * bheight * ref.sc) >> 1) + ((idx * bwidth) >> 1);
    mv_t mv = clip_mv(mv_arr[index], ypos, xpos, pic_w, pic_h, bwidth, bheight, sign);
    pred_luma<SP>(t, py + offpY, pstride, ry + offrY, ref.sy, bwidth, bheight, mv, sign, enable_bipred, pic_w, pic_h,
              xpos, ypos, bitdepth);
    if (luma_only) continue;   // the bi-prediction search only needs 2*org - luma prediction
    pred_chroma<SP>(t, pu + offpC, pstride >> 1, ru + offrC, ref.sc, bwidth >> 1, bheight >> 1, mv, sign, pic_w >> 1,
                pic_h >> 1, xc, yc, bitdepth);
    pred_chroma<SP>(t, pv + offpC, pstride >> 1, rv + offrC, ref.sc, bwidth >> 1, bheight >> 1, mv, sign, pic_w >> 1,
                pic_h >> 1, xc, yc, bitdepth);
  }
}

// average_blocks_all: truncating (a+b)>>1 (inter_prediction.c:228-247).
// Four 8-bit or two 16-bit samples per dword: (a + b) >> 1 per sample = (a & b) + (((a ^ b) >> 1) & M), M = all bits but each sample's
// top one (no carry crosses a sample).
template <int SP, typename PIX> TK_DEV void average_rows(const Team t, PIX* d_, const PIX* a_, const PIX* b_, int stride, int w, int h) {
  const int spp = 4 / (int)sizeof(PIX);
  const uint32_t M = sizeof(PIX) == 1 ? 0x7f7f7f7fu : 0x7fff7fffu;
  const int ppr = w / spp;
  for (int k = t.rank; k < ppr * h; k += t.size) {
    const int i = k / ppr, j = k - i * ppr;   // ppr is a power of two except on frame-edge rectangles
    const int o = i * stride + spp * j;
#if TK_HOST
    uint32_t x, y;
    __builtin_memcpy(&x, a_ + o, 4); __builtin_memcpy(&y, b_ + o, 4);
    const uint32_t r = (x & y) + (((x ^ y) >> 1) & M);
    __builtin_memcpy(d_ + o, &r, 4);
#else
    const uint32_t x = *(typename SpT<SP, const uint32_t>::ptr)(spc<SP>(a_) + o), y = *(typename SpT<SP, const uint32_t>::ptr)(spc<SP>(b_) + o);
    *(typename SpT<SP, uint32_t>::ptr)(spc<SP>(d_) + o) = (x & y) + (((x ^ y) >> 1) & M);
#endif
  }
}
template <int SP, typename PIX>
TK_DEV void average_yuv(const Team t, PIX* dy_, PIX* du_, PIX* dv_, const PIX* ay_, const PIX* au_, const PIX* av_,
                        const PIX* by_, const PIX* bu_, const PIX* bv_, int size, int bw, int bh) {
#ifndef TK_NOVEC
  // sample blocks are 16-byte aligned with a row pitch of `size` (>= 8): dword pieces whenever the width allows
  if (tk_uniform(!((bw | size) & 7))) {
    average_rows<SP>(t, dy_, ay_, by_, size, bw, bh);
    average_rows<SP>(t, du_, au_, bu_, size >> 1, bw >> 1, bh >> 1);
    average_rows<SP>(t, dv_, av_, bv_, size >> 1, bw >> 1, bh >> 1);
    return;
  }
#endif
  const auto dy = spc<SP>(dy_); const auto du = spc<SP>(du_); const auto dv = spc<SP>(dv_);
  const auto ay = spc<SP>(ay_); const auto au = spc<SP>(au_); const auto av = spc<SP>(av_);
  const auto by = spc<SP>(by_); const auto bu = spc<SP>(bu_); const auto bv = spc<SP>(bv_);
  for (int k = t.rank; k < bw * bh; k += t.size) {
    int i, j;
    split2(mk_div(bw), k, i, j);
    int o = i * size + j;
    dy[o] = (PIX)(((int)ay[o] + (int)by[o]) >> 1);
  }
  int cw = bw >> 1, ch = bh >> 1, cs = size >> 1;
  for (int k = t.rank; k < cw * ch; k += t.size) {
    int i, j;
    split2(mk_div(cw), k, i, j);
    int o = i * cs + j;
    du[o] = (PIX)(((int)au[o] + (int)bu[o]) >> 1);
    dv[o] = (PIX)(((int)av[o] + (int)bv[o]) >> 1);
  }
}

// ---------------------------------------------------------------------------------
// Intra
// ---------------------------------------------------------------------------------
template <typename PIX> struct IntraEdge {
  PIX left[2 * kMaxSb];
  PIX top[2 * kMaxSb];
  PIX top_left;
};

// make_top_and_left (intra_prediction.c:57-183).  rec_frame points at the CB's top-left sample in
// the reconstructed frame; rblock at the TU's top-left in the CB-local recon block (tb_split only);
// (i, j) = TU offset inside the CB; (ypos, xpos) = CB position in this plane.
// SP: address space of rblock (the coding block's own reconstruction buffer)
template <int SP, typename PIX>
TK_DEV void make_edges(const Team t, IntraEdge<PIX>* e, const PIX* rec_frame, int fstride, const PIX* rblock,
                       int rbstride, int i, int j, int ypos, int xpos, int size, int cb_upright, int cb_downleft,
                       int tb_split, int bitdepth) {
  e = tk_uniform_ptr(e); rec_frame = tk_uniform_ptr(rec_frame); rblock = tk_uniform_ptr(rblock); fstride = tk_uniform(fstride);
  rbstride = tk_uniform(rbstride); i = tk_uniform(i); j = tk_uniform(j); ypos = tk_uniform(ypos); xpos = tk_uniform(xpos); size = tk_uniform(size);
  cb_upright = tk_uniform(cb_upright); cb_downleft = tk_uniform(cb_downleft); tb_split = tk_uniform(tb_split); bitdepth = tk_uniform(bitdepth);
  const int len = 2 * size;
  const PIX dflt = (PIX)(128 << (bitdepth - 8));
  int downleft, upright;
  if (!tb_split) {
    downleft = cb_downleft;
    upright = cb_upright;
  } else {
    downleft = (j == 0 && (i == 0 || cb_downleft)) ? 1 : 0;
    upright = (j == 0 || (i == 0 && cb_upright)) ? 1 : 0;
  }
  const int leftlen = downleft ? size + 1 : size;
  const int toplen = upright ? size + 1 : size;
  const int top_from_block = tb_split && i != 0;
  const int left_from_block = tb_split && j != 0;
  const auto rblock_g = spc<SP>(rblock);  // CB-local recon block: per-wave scratch, in LDS for small CBs
  const TK_GLOBAL PIX* const frame_g = gptr(rec_frame);
  const auto e_top = ldsc(e->top);
  const auto e_left = ldsc(e->left);
  // tb_split==0 => i==j==0 so (rec_frame - fstride + j) is the reference's &rec_frame[-fstride+j].
  const int top_dflt = (ypos + i == 0);
  const int left_dflt = (xpos + j == 0);
  for (int k = t.rank; k < len; k += t.size) {
    PIX tv, lv;
    if (top_dflt) tv = dflt;
    else {
      const int kk = k < toplen ? k : toplen - 1;
      tv = top_from_block ? rblock_g[kk - rbstride] : frame_g[kk - fstride + j];
    }
    if (left_dflt) lv = dflt;
    else {
      int kk = k < leftlen ? k : leftlen - 1;
      lv = left_from_block ? rblock_g[kk * rbstride - 1] : frame_g[(i + kk) * fstride - 1];
    }
    e_top[k] = tv;
    e_left[k] = lv;
  }
  if (t.rank == 0) {
    PIX tl;
    if (top_dflt) {
      tl = left_dflt ? dflt : (left_from_block ? rblock_g[-1] : frame_g[i * fstride - 1]);  // = left[0]
    } else if (!top_from_block) {
      tl = xpos > 0 ? frame_g[-fstride + j - 1] : frame_g[-fstride + j];
    } else {
      tl = xpos > 0 ? (j > 0 ? rblock_g[-rbstride - 1] : frame_g[(i - 1) * fstride - 1]) : rblock_g[-rbstride];
    }
    *ldsc(&e->top_left) = tl;
  }
  t.sync();
}

template <typename PIX> TK_DEV int f121(const PIX* a, int k, int len) {
  int km = k > 0 ? k - 1 : 0, kp = k < len - 1 ? k + 1 : len - 1;
  return (a[km] + 2 * a[k] + a[kp] + 2) >> 2;
}
template <typename PIX> TK_DEV int f5(const PIX* a, int k, int size) {
  int k0 = clampi(k - 2, 0, size - 1), k1 = clampi(k - 1, 0, size - 1), k3 = clampi(k + 1, 0, size - 1),
      k4 = clampi(k + 2, 0, size - 1);
  return (int16_t)(a[k0] + 2 * a[k1] + 2 * a[k] + 2 * a[k3] + a[k4]);
}

// get_intra_prediction (intra_prediction.c:403-428) - writes size x size at dst (stride dstride).
// (ypos, xpos) = TU position in this plane (only the ==0 tests matter, for DC).
template <int SP, typename PIX>
TK_DEV void pred_intra(const Team t, const IntraEdge<PIX>* e, int ypos, int xpos, int size, PIX* dst_, int dstride,
                       int mode, int bitdepth) {
  PIX* dst = dst_;
  e = tk_uniform_ptr(e); dst = tk_uniform_ptr(dst); ypos = tk_uniform(ypos); xpos = tk_uniform(xpos); size = tk_uniform(size);
  dstride = tk_uniform(dstride); mode = tk_uniform(mode); bitdepth = tk_uniform(bitdepth);
  typedef TK_LDS PIX lpix;  // the edge arrays live in LDS on the device (see tk_common.h)
  const lpix* left = (const lpix*)ldsc(e->left);
  const lpix* top = (const lpix*)ldsc(e->top);
  const int tl = *ldsc(&e->top_left);
  const auto dsts = spc<SP>(dst);
  int dc = 0, tlF = 0, tlP = 0;
  if (mode == 0 || mode > 9) {
    const lpix* a = xpos != 0 ? left : top;
    const lpix* b = ypos != 0 ? top : left;
    unsigned sum = 0;
    for (int k = 0; k < size; k++) sum += (unsigned)a[k] + (unsigned)b[k];
    dc = (int)((sum + (unsigned)size) / (2u * (unsigned)size));
  } else if (mode == 1) {
    tlP = (int16_t)(left[1] + 2 * left[0] + 2 * tl + 2 * top[0] + top[1]);
  } else if (mode == 4 || mode == 7 || mode == 8) {
    tlF = (PIX)((2 * tl + left[0] + top[0] + 2) >> 2);
  }
  const Pow2 pw = mk_pow2(size);
  for (int k = t.rank; k < size * size; k += t.size) {
    int i, j;
    split2(pw, k, i, j);
    int v;
    switch (mode) {
      case 1:  // planar
        v = sat_pix((f5(left, i, size) + f5(top, j, size) - tlP + 4) / 8, bitdepth);
        break;
      case 2: v = left[i]; break;  // hor
      case 3: v = top[j]; break;   // ver
      case 4: {  // upleft
        int d = i - j;
        v = d > 0 ? f121(left, d - 1, size) : (d == 0 ? tlF : f121(top, -d - 1, size));
      } break;
      case 5: v = f121(top, i + j + 1, 2 * size); break;  // upright
      case 6: {  // upupright
        int d = i + 2 * j;
        v = (d & 1) ? f121(top, (d + 1) / 2, 2 * size) : ((f121(top, d / 2, 2 * size) + f121(top, d / 2 + 1, 2 * size)) >> 1);
      } break;
      case 7: {  // upupleft
        int d = i - 2 * j;
        if (d > 1) v = f121(left, d - 2, size);
        else if (d == 1) v = tlF;
        else if (d == 0) v = (tlF + f121(top, 0, size)) >> 1;
        else if (d & 1) v = f121(top, (-d) / 2, size);
        else v = (f121(top, (-d) / 2, size) + f121(top, (-d) / 2 - 1, size)) >> 1;
      } break;
      case 8: {  // upleftleft
        int d = 2 * i - j;
        if (d < -1) v = f121(top, -d - 2, size);
        else if (d == -1) v = tlF;
        else if (d == 0) v = (tlF + f121(left, 0, size)) >> 1;
        else if (d & 1) v = f121(left, d / 2, size);
        else v = (f121(left, d / 2, size) + f121(left, d / 2 - 1, size)) >> 1;
      } break;
      case 9: {  // downleftleft
        int d = 2 * i + j;
        v = (d & 1) ? f121(left, (d + 1) / 2, 2 * size) : ((f121(left, d / 2, 2 * size) + f121(left, d / 2 + 1, 2 * size)) >> 1);
      } break;
      default: v = dc; break;
    }
    dsts[i * dstride + j] = (PIX)v;
  }
}

}  // namespace tk
