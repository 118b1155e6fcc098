// Host-side exhaustive check of the border-class phases of a stride-1 transposed gather on a small map
// (csrc/conv_geom.h: build_transposed): for gathered extents 2..12, kernels 1..4 and every padding, each in-range
// (produced pixel, tap) pair must be issued by exactly one (phase, row, tap) triple with the right gathered coordinate.
//   hipcc --offload-arch=gfx950 -I include -I cross-modality-minipig-gan_amd/csrc tools/check_phase_classes.hip -o /tmp/c && /tmp/c
// (tests/test_host_logic.py builds and runs it.)
#include "conv_geom.h"
#include <cstdio>
#include <map>
#include <tuple>
using namespace mpgan;
namespace mpgan { void set_error(const char*, ...) {} }
int main() {
  int bad = 0, used = 0;
  for (int G = 2; G <= 12; ++G) for (int K = 1; K <= 4; ++K) for (int pd = 0; pd < K; ++pd) {
    int P = G + K - 1 - 2 * pd; if (P < 1) continue;
    int32_t gd[3] = {G, G + 1, G + 2}, pdh[3] = {P, P + 1, P + 2}, k[3] = {K, K, K}, s[3] = {1, 1, 1}, pad[3] = {pd, pd, pd};
    GatherConv p{};
    build_transposed(p, 1, gd, 64, pdh, 64, k, s, pad);
    if (p.nphase > 1) used++;
    // every (produced pixel, tap) pair with an in-range gathered coordinate must be issued exactly once
    std::map<std::tuple<int,int,int,int,int,int>, int> seen;
    for (int i = 0; i < p.nphase; ++i) {
      const Phase& ph = p.ph[i];
      for (int mz = 0; mz < ph.Mz; ++mz) for (int my = 0; my < ph.My; ++my) for (int mx = 0; mx < ph.Mx; ++mx)
        for (int jz = 0; jz < ph.nz; ++jz) for (int jy = 0; jy < ph.ny; ++jy) for (int jx = 0; jx < ph.nx; ++jx) {
          int oz = mz * p.ostride[0] + ph.oz, oy = my * p.ostride[1] + ph.oy, ox = mx * p.ostride[2] + ph.ox;
          int iz = mz * p.istride[0] + ph.dz0 + p.dstep[0] * jz, iy = my * p.istride[1] + ph.dy0 + p.dstep[1] * jy, ix = mx * p.istride[2] + ph.dx0 + p.dstep[2] * jx;
          int kz = ph.kz0 + p.kstep[0] * jz, ky = ph.ky0 + p.kstep[1] * jy, kx = ph.kx0 + p.kstep[2] * jx;
          if (oz >= pdh[0] || oy >= pdh[1] || ox >= pdh[2]) { bad++; continue; }
          if (iz < 0 || iz >= gd[0] || iy < 0 || iy >= gd[1] || ix < 0 || ix >= gd[2]) continue;   // masked by the kernels
          if (iz != oz + pd - kz || iy != oy + pd - ky || ix != ox + pd - kx) bad++;
          seen[{oz, oy, ox, kz, ky, kx}]++;
        }
    }
    if (p.classes) {
      // the packed work list (set_tile_grid + decode_block): every (phase, m-tile, n-tile, split) exactly once, rows dense
      for (int bm : {128, 256, 512}) {
        GatherConv q = p;
        q.N = 3;
        const long pairs = set_tile_grid(q, bm);
        q.ntiles = 2; q.ksplit = 2;
        if (pairs != phase_tile_rows(q, bm) || !q.packed) bad++;
        std::map<std::tuple<int,int,int,int>, int> hit;
        for (unsigned w = 0; w < (unsigned)(pairs * q.ntiles * q.ksplit); ++w) {
          const BlockId b = decode_block(q, w);
          const long tiles = ((long)q.N * q.ph[b.phase].Mz * q.ph[b.phase].My * q.ph[b.phase].Mx + bm - 1) / bm;
          if (b.phase < 0 || b.phase >= q.nphase || b.mt < 0 || b.mt >= tiles || b.nt < 0 || b.nt >= 2 || b.split < 0 || b.split >= 2 ||
              b.row != q.tile_start[b.phase] + b.mt || b.row >= pairs) bad++;
          hit[{b.phase, b.mt, b.nt, b.split}]++;
        }
        if ((long)hit.size() != pairs * 4) bad++;
      }
    }
    long want = 0;
    for (int oz = 0; oz < pdh[0]; ++oz) for (int oy = 0; oy < pdh[1]; ++oy) for (int ox = 0; ox < pdh[2]; ++ox)
      for (int kz = 0; kz < K; ++kz) for (int ky = 0; ky < K; ++ky) for (int kx = 0; kx < K; ++kx) {
        int iz = oz + pd - kz, iy = oy + pd - ky, ix = ox + pd - kx;
        if (iz < 0 || iz >= gd[0] || iy < 0 || iy >= gd[1] || ix < 0 || ix >= gd[2]) continue;
        want++;
        auto it = seen.find({oz, oy, ox, kz, ky, kx});
        if (it == seen.end() || it->second != 1) bad++;
      }
    if ((long)seen.size() != want) bad++;
  }
  printf("class-phase check: %d geometries used classes, %d errors, sizeof(GatherConv) %zu\n", used, bad, sizeof(GatherConv));
  return bad != 0;
}
