// kat_host_es.cpp - TEST INFRASTRUCTURE ONLY.  The 1-lane host build of the device functions early_skip_sub / early_skip_subC (thor_amd/csrc/tk_block.h: the two
// sub-block tests of check_early_skip - luma: 2x2 average + half-size transform against half the threshold; chroma: the SIMD form of calc_cbp the reference
// executes), so that tests/golden/kat7.npz (recorded from the reference's file-static check_early_skip_sub_block / _sub_blockC) pins them on the CPU
// (tests/test_kat_host.py).  On the device they run inside k_superblocks and are covered by the stream goldens.
#include "../../thor_amd/csrc/tk_block.h"
#include "../../thor_amd/csrc/tk_tables.h"
namespace tk { Tables g_tab; long long g_prune_stat[8]; }
using namespace tk;
extern "C" int h_early_skip_sub(int chroma, const uint8_t* org, int ostride, const uint8_t* pred, int pstride, int size, int qp, float thr, int bitdepth) {
  static SmallWs<uint8_t> sws;
  static WgShared sh;
  static bool inited = false;
  if (!inited) { init_tables(&g_tab); xform_tables_fill(&sh.tabs, 0, 1); inited = true; }   // the constant tables + the workgroup's transposed bases
  static BigWs<uint8_t> big;
  static FrameJob<uint8_t> J;
  J.cfg.bitdepth = bitdepth;
  TeamWs<uint8_t> ws = make_ws(&sws, &sh, &big);
  const Team t = mk_team(0, 1);
  return chroma ? early_skip_subC<uint8_t, SP_GLOBAL>(t, J, &ws, org, ostride, pred, pstride, size, qp, thr)
                : early_skip_sub<uint8_t, SP_GLOBAL>(t, J, &ws, org, ostride, pred, pstride, size, qp, thr);
}
