// Diagnostic (not part of the product): in-kernel clock and cycles per K/V tile of the cross-attention
// forward main loop (MI355X_MICROARCH.md "DVFS give-back" item 6: d(s_memtime)/d(s_memrealtime) x 100 MHz,
// after >= 2 s of back-to-back launches on random data).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DPETR_DIAG_CLOCK -x hip scripts/diag_clock.cpp petr_amd/csrc/api.cpp -o scripts/bin/diag_clock
#include "../petr_amd/csrc/mha_fwd.hip"
#include <algorithm>
#include <chrono>
#include <map>
#include <random>
#include <vector>

int main(int argc, char** argv) {
  const int B = 1, H = 8, Q = 900, L = argc > 1 ? atoi(argv[1]) : 4224, ns = argc > 2 ? atoi(argv[2]) : 8;
  std::mt19937 rng(0);
  std::normal_distribution<float> nd;
  auto mk = [&](size_t n) {
    std::vector<float> h(n);
    for (auto& x : h) x = nd(rng);
    float* d;
    hipMalloc(&d, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
  };
  float *q = mk((size_t)B * H * Q * 32), *k = mk((size_t)B * H * L * 32), *v = mk((size_t)B * H * L * 32), *o, *lse;
  hipMalloc(&o, (size_t)B * H * Q * 32 * 4);
  hipMalloc(&lse, (size_t)B * H * Q * 4);
  petr_mha_fwd_args a{};
  a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse;
  a.B = B; a.H = H; a.Q = Q; a.L = L;
  a.q_bs = (long)H * Q * 32; a.q_hs = (long)Q * 32; a.q_rs = 32;
  a.k_bs = (long)H * L * 32; a.k_hs = (long)L * 32; a.k_rs = 32;
  a.v_bs = a.k_bs; a.v_hs = a.k_hs; a.v_rs = 32;
  a.o_bs = a.q_bs; a.o_hs = a.q_hs; a.o_rs = 32;
  a.scale = 0.17677669f; a.n_split = ns;
  a.ws_bytes = petr_mha_fwd_workspace_bytes(B, H, Q, L, ns);
  hipMalloc(&a.ws, a.ws_bytes + 16);
  if (!getenv("DIAG_STATIC")) {
    hipMalloc(&a.sched, 4096);
    hipMemset(a.sched, 0, 4096);
  }
  auto t0 = std::chrono::steady_clock::now();
  long n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
    for (int i = 0; i < 200; ++i)
      if (petr_mha_fwd(&a, nullptr)) { printf("error: %s\n", petr_last_error()); return 1; }
    hipDeviceSynchronize();
    n += 200;
  }
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::vector<unsigned long long> d(16 * 4096);
  hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(g_diag), d.size() * 8);
  const int wgs = ((Q + 127) / 128) * B * H * ns;
  std::vector<double> clk, cyc_tile;
  std::map<unsigned, int> per_simd, per_cu;   // resident waves of the LAST launch per (xcc,se,cu,simd) / per cu
  std::map<unsigned, double> cu_cyc;
  std::map<int, int> hist;
  std::map<int, std::pair<double, int>> by_qb, by_split, by_xcc, by_bh;
  for (int i = 0; i < wgs && i < 4096; ++i) {
    for (int w = 0; w < 4; ++w) {
      const unsigned long long* e = &d[4 * (4 * i + w)];
      const unsigned tiles = (unsigned)e[2] & 0xffff;
      const unsigned wid = (unsigned)(e[2] >> 32);
      if (!e[1] || !tiles) continue;
      const unsigned hw = (unsigned)e[3], xcc = (unsigned)(e[3] >> 32) & 15;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      const unsigned cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu;
      per_simd[cukey * 4 + simd]++;
      if (w == 0) { per_cu[cukey]++; cu_cyc[cukey] = std::max(cu_cyc[cukey], (double)e[0] / tiles); }
      if (w == 0 && tiles >= 8) {
        const int nqb = (Q + 127) / 128;
        const int qb = wid % nqb, split = (wid / nqb) % ns, bh = wid / nqb / ns;
        const double cpt = (double)e[0] / tiles;
        hist[(int)(cpt / 500)]++;
        by_qb[qb].first += cpt; by_qb[qb].second++;
        by_split[split].first += cpt; by_split[split].second++;
        by_xcc[xcc].first += cpt; by_xcc[xcc].second++;
        by_bh[bh].first += cpt; by_bh[bh].second++;
        if (getenv("DIAG_DUMP")) printf("wg %4d w %4d xcc %u se %u sh %u cu %2u simd %u qb %d split %d bh %d tiles %u cyc/tile %.0f\n", i, wid, xcc, se, sh, cu, simd, qb, split, bh, tiles, cpt);
      }
      if (w == 0) {
        clk.push_back((double)e[0] / (double)e[1] * 100.0);
        if (tiles >= 8) cyc_tile.push_back((double)e[0] / tiles);
      }
    }
  }
  std::map<int, int> hs, hc;
  for (auto& kv : per_simd) hs[kv.second]++;
  for (auto& kv : per_cu) hc[kv.second]++;
  printf("CUs seen: %zu; workgroups per CU histogram:", per_cu.size());
  for (auto& kv : hc) printf("  %d WG: %d CUs", kv.first, kv.second);
  printf("\nwaves per SIMD histogram:");
  for (auto& kv : hs) printf("  %d waves: %d SIMDs", kv.first, kv.second);
  printf("\n");
  {
    std::map<int, std::pair<double, int>> m;
    for (auto& kv : per_cu) { m[kv.second].first += cu_cyc[kv.first]; m[kv.second].second++; }
    for (auto& kv : m) printf("  CUs with %d WGs: mean max-cycles-per-tile %.0f\n", kv.first, kv.second.first / kv.second.second);
  }
  std::sort(clk.begin(), clk.end());
  std::sort(cyc_tile.begin(), cyc_tile.end());
  printf("cycles-per-tile histogram (500-cycle bins, WGs with >= 8 tiles):");
  for (auto& kv : hist) printf("  %d:%d", kv.first * 500, kv.second);
  auto show = [&](const char* n, std::map<int, std::pair<double, int>>& m) {
    printf("\nmean cycles/tile by %s:", n);
    for (auto& kv : m) printf("  %d:%.0f", kv.first, kv.second.first / kv.second.second);
  };
  show("query block", by_qb); show("split", by_split); show("xcc", by_xcc); show("head", by_bh);
  printf("\n");
  printf("L=%d ns=%d: %.1f us per call (fwd+combine, back-to-back), %ld calls\n", L, ns, wall / n * 1e6, n);
  printf("in-kernel clock MHz: median %.0f  (min %.0f max %.0f) over %zu workgroups\n", clk[clk.size() / 2], clk.front(),
         clk.back(), clk.size());
  if (!cyc_tile.empty())
    printf("loop cycles per 64-key tile (wave 0 of each WG): median %.0f min %.0f max %.0f; MFMA-only bound per wave 4096, "
           "two waves per SIMD 8192\n", cyc_tile[cyc_tile.size() / 2], cyc_tile.front(), cyc_tile.back());
  return 0;
}
