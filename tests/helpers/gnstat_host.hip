// CPU harness for tests/test_host_gn_stats.py: runs the GroupNorm pair-statistics tree of h-edit_amd/csrc/gnstat.h -- the very
// helper functions the igemm epilogue and the split-K reduce call, compiled for the host -- in the thread layouts of the three
// producer forms, one emulated thread after the other:
//   form 0: 128-row tile, 256 threads (igemm_kernel<128, 128, ...>, splitk_reduce_gn_kernel): piece c of rows r + 16 it
//   form 1: 256-row tile, 512 threads (igemm_kernel<256, 128, ...>):                          piece c of rows r + 32 it
// Input: a 256 x 128 tile of bf16 bits (argv[1], 65536 bytes).  Output (argv[2]): float32 [form][unit 2][pair 64][2].
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../h-edit_amd/csrc/gnstat.h"

template <int BM>
static void run_tile(const uint16_t* tile /* [BM][128] */, float* out /* [BM / 128][64][2] */) {
  constexpr int NT = BM * 2, CHUNKS = 16, ITER = 8, RSTEP = NT / CHUNKS, UNITS = BM / GNS_UNIT;
  std::vector<float2> red((size_t)UNITS * 32 * 64);
  for (int tid = 0; tid < NT; ++tid) {
    const int r = tid / CHUNKS, c = tid - r * CHUNKS;
    GnPiece t[2];
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * NT;                  // the store mapping of the epilogue
      const int ml = idx / CHUNKS, cc = idx - ml * CHUNKS;
      if (cc != c || ml != r + RSTEP * it) { fprintf(stderr, "mapping\n"); exit(2); }
      const uint32_t* w = reinterpret_cast<const uint32_t*>(tile + (size_t)ml * 128 + c * 8);
      const GnPiece g = gns_piece(w[0], w[1], w[2], w[3]);
      if (gns_first<RSTEP>(it)) t[gns_slot<RSTEP>(it)] = g; else gns_add(t[gns_slot<RSTEP>(it)], g);
    }
    gns_store_t(red.data(), gns_red_row<RSTEP>(r, 0), c, t[0]);
    gns_store_t(red.data(), gns_red_row<RSTEP>(r, 1), c, t[1]);
  }
  for (int tid = 0; tid < UNITS * 64; ++tid) {
    const float2 v = gns_fold_unit(red.data(), tid >> 6, tid & 63);
    out[(size_t)tid * 2] = v.x;
    out[(size_t)tid * 2 + 1] = v.y;
  }
}

int main(int argc, char** argv) {
  if (argc != 3) return 1;
  std::vector<uint16_t> tile(256 * 128);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(tile.data(), 2, tile.size(), f) != tile.size()) return 1;
  fclose(f);
  std::vector<float> out(2 * 2 * 64 * 2);
  run_tile<128>(tile.data(), out.data());                       // form 0, unit 0
  run_tile<128>(tile.data() + 128 * 128, out.data() + 128);     // form 0, unit 1
  run_tile<256>(tile.data(), out.data() + 256);                 // form 1, both units
  f = fopen(argv[2], "wb");
  if (!f || fwrite(out.data(), 4, out.size(), f) != out.size()) return 1;
  fclose(f);
  return 0;
}
