// Host-only check (no GPU needed) of the process's polling host threads (spartan2_amd/csrc/walk_pool.hpp): WalkPool::run is a parallel-for in which every
// part runs exactly once whoever claims it (walkers awake, asleep, or none at all), regions may follow each other without a pause and may be posted
// from several owner threads at once (eight contexts share the pool), and a table walk (post / finish) adds up the entries it was given. Built and run by
// tests/test_walk_pool_cpu.py: `pool_check [walkers-hot 0|1]`, SPARTAN_WALKERS from the environment.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../spartan2_amd/csrc/walk_pool.hpp"

#if !defined(__HIP_DEVICE_COMPILE__)
struct Job {
  std::atomic<unsigned> hits[32];
  std::atomic<unsigned long long> sum{0};
  const unsigned* data;
  size_t n;
};
static void part(void* arg, unsigned p, unsigned np) {
  Job& j = *static_cast<Job*>(arg);
  j.hits[p].fetch_add(1);
  unsigned long long s = 0;
  for (size_t i = j.n * p / np; i < j.n * (p + 1) / np; ++i) s += j.data[i];
  j.sum.fetch_add(s);
}
static int regions(int rounds, bool hot, unsigned seed) {
  sp::WalkPool& pool = sp::WalkPool::get();
  std::vector<unsigned> data(4096);
  unsigned long long want = 0;
  for (size_t i = 0; i < data.size(); ++i) {
    data[i] = (unsigned)(i * 2654435761u + seed);
    want += data[i];
  }
  int bad = 0;
  for (int r = 0; r < rounds; ++r) {
    if (hot && (r & 63) == 0) pool.keep_hot(2000);
    Job j;
    for (auto& h : j.hits) h.store(0);
    j.data = data.data();
    j.n = data.size();
    const unsigned np = 1 + (unsigned)((r * 7 + seed) % 32);
    pool.run(np, part, &j);
    for (unsigned p = 0; p < 32; ++p) bad += j.hits[p].load() != (p < np ? 1u : 0u);
    bad += j.sum.load() != want;
  }
  return bad;
}
#endif
int main(int argc, char** argv) {
#if !defined(__HIP_DEVICE_COMPILE__)
  const bool hot = argc > 1 && atoi(argv[1]) != 0;
  int bad = regions(2000, hot, 1);
  // several owners at once (more of them than the pool has batch slots: the ones that find none run their parts themselves)
  std::vector<std::thread> owners;
  std::atomic<int> bad_mt{0};
  for (int t = 0; t < 12; ++t) owners.emplace_back([&, t] { bad_mt.fetch_add(regions(400, hot, 100 + (unsigned)t)); });
  for (auto& t : owners) t.join();
  bad += bad_mt.load();
  // a table walk: n copies of one affine point, cut into parts, add up to n * P
  {
    sp::WalkPool& pool = sp::WalkPool::get();
    aff_t g;  // any point of the curve: the first x = 1, 2, ... whose x^3 - 3x + b is a square (p = 3 mod 4: the root is a (p + 1) / 4-th power)
    {
      uint32_t e[8], c = 0;
      for (int i = 0; i < 8; ++i) e[i] = sp_addc(FpP::P(i), i == 0 ? 1u : 0u, c);
      for (int i = 0; i < 8; ++i) e[i] = (e[i] >> 2) | (i < 7 ? e[i + 1] << 30 : c << 30);
      for (uint64_t x = 1;; ++x) {
        g.x = fe_from_u64<B>(x);
        const fe_t rhs = fe_add<B>(fe_sub<B>(fe_mul<B>(fe_sqr<B>(g.x), g.x), fe_add<B>(fe_dbl<B>(g.x), g.x)), T256::b());
        g.y = fe_pow<B>(rhs, e);
        if (aff_on_curve(g)) break;
      }
    }
    for (unsigned n : {1u, 2u, 7u, 33u, 200u}) {
      if (hot) pool.keep_hot(2000);
      sp::WalkPool::Batch* b = pool.acquire();
      if (!b) {
        ++bad;
        continue;
      }
      for (unsigned i = 0; i < n; ++i) b->ents[i] = &g;
      b->n_ents = n;
      pool.post(b, 9);
      const xyzz_t got = pool.finish(b);
      jac_t want = jac_identity();
      for (unsigned i = 0; i < n; ++i) want = jac_add_mixed(want, g);
      const aff_t a1 = jac_to_affine(xyzz_to_jac(got)), a2 = jac_to_affine(want);
      bad += !(fe_eq(a1.x, a2.x) && fe_eq(a1.y, a2.y));
    }
  }
  // a region of idempotent parts with a straggler: whoever runs part 1 FIRST sleeps 20 ms inside it (a walker that has lost its core); the owner's
  // collect_fn must not wait for it (it runs the part again), the results must be complete, and the data's reference must be dropped exactly once -
  // by the owner when nobody is late, by the straggler otherwise
  {
    struct Shared {
      std::atomic<int> first{0}, refs{0}, dropped{0};
      std::atomic<unsigned> out[8];
    };
    sp::WalkPool& pool = sp::WalkPool::get();
    for (int rep = 0; rep < 6; ++rep) {
      Shared* S = new Shared();
      for (auto& o : S->out) o.store(0);
      if (hot) pool.keep_hot(50000);
      auto fn = [](void* a, unsigned p, unsigned) {
        Shared* s = static_cast<Shared*>(a);
        if (p == 1 && s->first.fetch_add(1) == 0) std::this_thread::sleep_for(std::chrono::milliseconds(20));
        s->out[p].store(100 + p);
      };
      auto drop = [](void* a) {
        Shared* s = static_cast<Shared*>(a);
        s->dropped.fetch_add(1);
      };
      sp::WalkPool::Batch* b = pool.post_fn(8, fn, S);
      if (!b) {  // no walkers: nothing to test
        delete S;
        break;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));  // let the walkers claim (the sleeper among them, when there are walkers awake)
      const auto t0 = std::chrono::steady_clock::now();
      const bool late = pool.collect_fn(b, 200000);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      for (unsigned p = 0; p < 8; ++p) bad += S->out[p].load() != 100 + p;
      // a walker slept inside part 1: the owner came back within a few ms without it; the owner itself slept (it claimed part 1 first): no straggler
      if (late && ms > 15.0) ++bad;
      pool.end_fn(b, drop, S);
      std::this_thread::sleep_for(std::chrono::milliseconds(40));  // the straggler has finished by now
      bad += S->dropped.load() != 1;
      delete S;
    }
  }
  printf("walk pool (%d walkers, %s): %d mismatches\n", sp::WalkPool::get().walkers(), hot ? "polling" : "asleep", bad);
  return bad ? 1 : 0;
#else
  return 0;
#endif
}
