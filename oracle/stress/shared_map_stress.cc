// TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
//
// Stress harness for variant (ii) of the CPU baseline (ONE reference cuckoohash_map
// shared by P threads, oracle/ref_driver.cc ref_ps_new_shared / ref_ps_step): the table starts at
// capacity 1 and grows through every doubling form of the reference map (full rehash below
// kMaxNumLocks buckets, lazy per-lock migration above) while P threads upsert.  Linked against
// ref_driver.cc directly (oracle/Makefile target `stress`), plain and with -fsanitize=address /
// thread; a SIGSEGV handler prints the faulting thread's stack.  Round 4's driver line lost the
// variant to an rc -11 on the 256-core GPU box; this is the tool that looks for it.
//
//   shared_map_stress <threads> <steps> [zipf|uniform] [initial_capacity]
//
// What it found (round 5, 256-core GPU box, profiles/r05/cpu_baseline_rc11.md): from capacity 1, 2 of 12
// runs die in the FIRST step with a null `buckets_` pointer under a bucket lock whose hashpower check
// had passed (try_find_insert_bucket, cuckoohash_map.hpp:1405; slot_search, :1741).  The window is the
// reference map's own (upstream libcuckoo's cuckoo_fast_double, cuckoohash_map.hpp:1814-1815):
// `old_buckets_.swap(buckets_)` makes hashpower() report the PREVIOUS table's power again, with a null
// bucket pointer, until the move-assignment on the next line; a thread that slept through a whole
// doubling on a lock of a locks array the doubler no longer takes (arrays are replaced while the table
// has fewer than kMaxNumLocks = 65 536 buckets, :1915-1935) passes check_hashpower in that window.
// From >= 2^18 slots the locks array is at its final size from the start and the window cannot open.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

extern "C" {
void* ref_ps_new_shared(int P, int dim, int opt, float init_acc, float wd, float init_value, uint64_t cap);
int64_t ref_ps_step(void* h, const int64_t* ids, int64_t n, const float* grads, float lr, int64_t update_time,
                    float* emb_out);
int64_t ref_ps_size(void* h);
int64_t ref_ps_lookup(void* h, const int64_t* ids, int64_t n, float* out);
void ref_ps_free(void*);
}

static void on_segv(int sig, siginfo_t* si, void*) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  char msg[128];
  const int len = snprintf(msg, sizeof(msg), "signal %d at address %p, stack:\n", sig, si->si_addr);
  (void)!write(2, msg, size_t(len));
  backtrace_symbols_fd(frames, n, 2);
  _exit(139);
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 256;
  const int steps = argc > 2 ? atoi(argv[2]) : 40;
  const bool uniform = argc > 3 && !strcmp(argv[3], "uniform");
  const uint64_t cap0 = argc > 4 ? strtoull(argv[4], nullptr, 10) : 1;
  setvbuf(stdout, nullptr, _IOLBF, 0);   // (a crash must not take the progress lines with it)
  struct sigaction sa {};
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_SIGINFO;
  sigaction(SIGSEGV, &sa, nullptr);
  sigaction(SIGBUS, &sa, nullptr);
  const int B = 65536, D = 64;
  void* ps = ref_ps_new_shared(P, D, /*adagrad*/ 1, 0.1f, 0.f, 0.f, cap0);
  std::mt19937_64 rng(1);
  std::vector<int64_t> ids(B);
  std::vector<float> g(size_t(B) * D, 0.01f), emb(size_t(B) * D);
  int64_t distinct_total = 0;
  for (int s = 0; s < steps; ++s) {
    for (auto& x : ids) {
      if (uniform) {
        x = int64_t(rng() >> 16);
      } else {   // a heavy-tailed rank (exponent ~1.2), spread over 48 bits
        const double u = (rng() >> 11) * (1.0 / 9007199254740992.0);
        x = int64_t(uint64_t(1.0 / pow(1.0 - u * 0.999999, 5.0)) * 0x9E3779B97F4A7C15ull % (1ull << 48));
      }
    }
    const int64_t U = ref_ps_step(ps, ids.data(), B, g.data(), 0.001f, 1700000000 + s, emb.data());
    distinct_total += U;
    if (s % 5 == 0 || s == steps - 1)
      printf("step %d distinct %ld rows %ld\n", s, long(U), long(ref_ps_size(ps)));
  }
  // every id of the last batch must be found (a lost key after a doubling would show here)
  std::vector<float> out(size_t(B) * D);
  const int64_t found = ref_ps_lookup(ps, ids.data(), B, out.data());
  printf("found %ld of %d after %d steps\n", long(found), B, steps);
  ref_ps_free(ps);
  return found == B ? 0 : 2;
}
