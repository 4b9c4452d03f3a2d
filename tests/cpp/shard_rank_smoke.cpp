// Drives sfmloc::ShardRank (include/sfmloc_engine.hpp) as a C++ host of a sharded run would, world size 1: the whole map
// is the one shard, the "all-gathers" are the identity.  Two copies of a query go through stage 1 (a gang session) and
// stage 2 (merge contexts); both results must be the unsharded sfmloc_localize's, bit for bit.
//   shard_rank_smoke <sfmDataDir> <matchDir> <query.desc> <query.feat> <width> <height>      -> prints OK <inliers>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/sfmloc_engine.hpp"

int main(int argc, char **argv) {
  if (argc < 7) return 2;
  try {
    sfmloc::LocalizeEngine eng(argv[1], argv[2], "", 0.6, 25, 4.0, false, 0, 0);
    FILE *f = fopen(argv[3], "rb");
    if (!f) return 3;
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 8, SEEK_SET);
    const uint32_t n = (uint32_t)((bytes - 8) / 64);
    std::vector<uint8_t> desc((size_t)n * 64);
    if (fread(desc.data(), 1, desc.size(), f) != desc.size()) return 3;
    fclose(f);
    std::vector<float> xy;
    f = fopen(argv[4], "r");
    double x, y, s, a;
    while (f && fscanf(f, "%lf %lf %lf %lf", &x, &y, &s, &a) == 4) {
      xy.push_back((float)x);
      xy.push_back((float)y);
    }
    if (f) fclose(f);
    if (xy.size() != (size_t)n * 2) return 3;
    sfmloc_map *map = eng.map();
    sfmloc_query *q[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; ++k)
      if (sfmloc_query_create(map, desc.data(), xy.data(), n, (uint32_t)atoi(argv[5]), (uint32_t)atoi(argv[6]), &q[k]))
        throw std::runtime_error(sfmloc_last_error());
    sfmloc_pose ref;
    std::vector<uint32_t> rq(65536), rl(65536);
    if (sfmloc_localize(map, q[0], nullptr, 0, &ref, rq.data(), rl.data(), 65536)) throw std::runtime_error(sfmloc_last_error());

    const uint32_t budget = 512;
    void *packed = nullptr;
    if (hipMalloc(&packed, sfmloc::ShardRank::packedBytes(2, budget)) != hipSuccess) return 4;
    sfmloc::ShardRank rank(map, 4, 2, 2);
    rank.stage1(q, 2, packed, budget);                 // both queries in one gang session of two contexts
    for (uint32_t i = 0; i < 2; ++i) {                 // (a real host all-gathers `packed` here, ordered by events)
      if (sfmloc_context_signal(rank.context(i), nullptr)) throw std::runtime_error(sfmloc_last_error());
    }
    if (sfmloc_context_wait(rank.mergeContext(0), nullptr)) throw std::runtime_error(sfmloc_last_error());
    const uint32_t idx[2] = {0, 1};
    rank.stage2(q, idx, 2, packed, 1, 0, 2, budget);
    for (uint32_t k = 0; k < 2; ++k) {
      sfmloc_pose p;
      std::vector<uint32_t> pq(65536), pl(65536);
      rank.finish(k, &p, pq.data(), pl.data(), 65536);
      if (p.ok != ref.ok || p.n_inliers != ref.n_inliers || memcmp(p.P, ref.P, sizeof(p.P)) ||
          memcmp(pq.data(), rq.data(), sizeof(uint32_t) * (size_t)ref.n_inliers) ||
          memcmp(pl.data(), rl.data(), sizeof(uint32_t) * (size_t)ref.n_inliers)) {
        printf("MISMATCH query %u: ok %d/%d inliers %d/%d\n", k, p.ok, ref.ok, p.n_inliers, ref.n_inliers);
        return 1;
      }
    }
    printf("OK %d\n", ref.n_inliers);
    (void)hipFree(packed);
    for (int k = 0; k < 2; ++k) sfmloc_query_destroy(q[k]);
  } catch (const std::exception &e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
