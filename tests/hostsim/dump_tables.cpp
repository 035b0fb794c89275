// TEST INFRASTRUCTURE: prints the engine's generated constant tables as JSON (tests/test_tables.py).
#include <cstdio>
#include "../../thor_amd/csrc/tk_tables.h"
namespace tk { Tables g_tab; }
template <typename T> static void dump(const char* n, const T* a, int c, bool last = false) {
  printf("\"%s\": [", n);
  for (int i = 0; i < c; i++) printf("%s%d", i ? "," : "", (int)a[i]);
  printf("]%s\n", last ? "" : ",");
}
int main() {
  tk::Tables t;
  tk::init_tables(&t);
  printf("{\n");
  dump("zigzag16", t.zz4, 16); dump("zigzag64", t.zz8, 64); dump("zigzag256", t.zz16, 256);
  dump("chroma_qp", t.chroma_qp, 52); dump("dct4", t.dct4, 16); dump("dct8", t.dct8, 64); dump("dct16", t.dct16, 256);
  dump("dct32", t.dct32, 1024); dump("beta", t.beta, 52); dump("tc", t.tc, 56);
  printf("\"lambda\": [");
  for (int i = 0; i < 52; i++) printf("%s%.4f", i ? "," : "", tk::kSquaredLambdaQP[i]);
  printf("],\n");
  int gq[6], gd[6];
  for (int i = 0; i < 6; i++) { gq[i] = 0; gd[i] = 0; }
  printf("\"izz_check\": %d\n}\n", t.izz16[t.zz16[37]] == 37);
  return 0;
}
