// multi_device.cpp -- one process, one host thread + one stream + its own plans per GPU, over
// the C-ABI of include/hexl_amd.h and nothing else (no HIP header, no torch): what a C++ caller
// of the reference (SEAL, OpenFHE) that wants more than one GPU writes.
//
// The job is BASELINE.json configs[3] -- N = 65536, 8 RNS primes x 4096 polynomials, in-place
// ForwardNTT(1,1) + InverseNTT(1,1) -- or a smaller one of the same shape (--n, --batch,
// --primes).  The flat (prime, polynomial) index is cut into contiguous shards, one per device
// (the rule of hexl_amd/sharding.py: job_partition -- shard g =
// [g U / G, (g + 1) U / G)); each prime's slice of a shard is transformed with that prime's
// plan, whole primes through hexl_amd_ntt_forward_rns.  There is no exchange between devices:
// the per-modulus loops of hexl/experimental/seal/key-switch-internal.cpp:51-90 are independent
// units.  "--scaling weak" gives every device its own prime x batch instead (the job grows).
//
// Worker threads never set a current device: plans carry theirs, device memory is allocated
// for an explicit device, and every other entry point runs on the device that owns the stream
// it is handed (include/hexl_amd.h, "Devices and streams").
//
// Verification: after the first forward pass the first and the last polynomial of every
// shard are bit-compared with the same polynomial transformed alone on device --ref-device
// (default: the first device of the list) by the main thread; after the inverse pass with the
// generator's input (round trip).  Timing: W warm-up steps, then K steps between two barriers;
// the job's time is the slowest thread's.  One JSON line on stdout.
//
// Round 6: before the warm-up every worker also proves that ITS plans on ITS device compute the
// right thing.  "--probe q:fwd_sha256:inv_sha256" (one per modulus; bench.py passes the digests
// of tests/golden/ntt_definition_fixtures.json, pinned to the big-integer definition of the
// transform) makes each worker transform splitmix64(seed 1) mod q forward and splitmix64(seed
// 1001) mod q inverse with each of its plans, on its device, and compare the SHA-256 of the
// results; hexl_amd_ntt_device(plan) must be the worker's device.  The line carries
// per_rank_probe_ok and per_rank_plan_device.
//
//   multi_device [--devices 0,1,...|all] [--scaling strong|weak] [--n N] [--batch B]
//                [--primes P] [--bits 54] [--steps K] [--warmup W] [--probe q:hex:hex ...]
// "--devices 0,0" runs two worker threads on one GPU (the dry run of the N > 1 code on a
// one-GPU box).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hexl_amd.h"

namespace {

struct Segment {
  uint64_t prime, first, count;
};

// hexl_amd/sharding.py: shard_range + units_by_prime
std::vector<Segment> shard_segments(uint64_t total, uint64_t per_prime, uint64_t world, uint64_t rank) {
  const uint64_t begin = rank * total / world, end = (rank + 1) * total / world;
  std::vector<Segment> out;
  for (uint64_t u = begin; u < end;) {
    const uint64_t prime = u / per_prime, first = u % per_prime;
    const uint64_t count = std::min(end - u, per_prime - first);
    out.push_back({prime, first, count});
    u += count;
  }
  return out;
}

class Barrier {  // C++17: no std::barrier
 public:
  explicit Barrier(int n) : n_(n) {}
  void wait() {
    std::unique_lock<std::mutex> lock(mu_);
    const uint64_t gen = gen_;
    if (++arrived_ == n_) {
      arrived_ = 0;
      ++gen_;
      cv_.notify_all();
    } else {
      cv_.wait(lock, [&] { return gen_ != gen; });
    }
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int n_, arrived_ = 0;
  uint64_t gen_ = 0;
};

// SHA-256 (FIPS 180-4), for the per-worker probe.
struct Sha256 {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void block(const uint8_t* p) {
    static const uint32_t k[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
        0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i)
      w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + k[i] + w[i];
      const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  // digest of `bytes` bytes (a multiple of 64 here: whole polynomials), as lower-case hex
  static std::string hex(const void* data, size_t bytes) {
    Sha256 s;
    const uint8_t* p = (const uint8_t*)data;
    size_t off = 0;
    for (; off + 64 <= bytes; off += 64) s.block(p + off);
    uint8_t tail[128] = {0};
    const size_t rem = bytes - off;
    memcpy(tail, p + off, rem);
    tail[rem] = 0x80;
    const size_t tl = rem + 9 <= 64 ? 64 : 128;
    const uint64_t bits = (uint64_t)bytes * 8;
    for (int i = 0; i < 8; ++i) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    s.block(tail);
    if (tl == 128) s.block(tail + 64);
    char out[65];
    for (int i = 0; i < 8; ++i) snprintf(out + 8 * i, 9, "%08x", s.h[i]);
    return std::string(out, 64);
  }
};

struct ProbeDigest {
  uint64_t q;
  std::string fwd, inv;
};

struct Config {
  std::vector<ProbeDigest> probes;
  std::vector<int> devices;
  bool weak = false;
  uint64_t n = 65536, batch = 4096, primes = 8, bits = 54;
  int steps = 5, warmup = 2, ref_device = -1;
};

struct Worker {
  int index = 0, device = 0;
  std::vector<Segment> segments;
  uint64_t polys = 0;
  double seconds = 0;   // its own K steps
  std::string error;    // empty: fine
  int probe_ok = -1;    // 1 / 0: the digests of its plans' probe transforms matched / did not; -1: no probe asked
  int plan_device = -1; // hexl_amd_ntt_device of its plans (-2: they disagree)
  // first / last polynomial after the first forward pass and after the round trip
  std::vector<uint64_t> fwd_first, fwd_last, back_first, back_last;
};

#define CK(call)                                                                  \
  do {                                                                            \
    if ((call) != HEXL_AMD_OK) {                                                  \
      w.error = std::string(#call) + ": " + hexl_amd_last_error();                \
      return;                                                                     \
    }                                                                             \
  } while (0)

void run_worker(Worker& w, const Config& cfg, const std::vector<uint64_t>& moduli, Barrier& bar) {
  const uint64_t n = cfg.n;
  void* stream = nullptr;
  uint64_t* data = nullptr;
  std::vector<hexl_amd_ntt*> plans;
  bool passed_first = false;  // this worker has met the others at the start of the timed region
  auto body = [&]() {
    CK(hexl_amd_stream_create(&stream, w.device));
    for (const Segment& s : w.segments) {
      hexl_amd_ntt* p = nullptr;
      CK(hexl_amd_ntt_create(&p, n, moduli[s.prime % moduli.size()], 0, w.device));
      plans.push_back(p);
    }
    CK(hexl_amd_device_alloc((void**)&data, w.polys * n * sizeof(uint64_t), w.device));
    std::vector<uint64_t*> seg_ptr;
    {
      uint64_t off = 0;
      for (size_t i = 0; i < w.segments.size(); ++i) {
        const Segment& s = w.segments[i];
        seg_ptr.push_back(data + off * n);
        // polynomial (prime, poly) always gets seed 1 + prime * batch + poly, whoever owns it
        CK(hexl_amd_fill_splitmix(seg_ptr[i], n, s.count, 1 + s.prime * cfg.batch + s.first,
                                  moduli[s.prime % moduli.size()], stream));
        off += s.count;
      }
    }
    // the probe: every plan of this worker, on this worker's device, against the definition digests
    for (size_t i = 0; i < plans.size(); ++i) {
      const int pd = hexl_amd_ntt_device(plans[i]);
      w.plan_device = (i == 0 || w.plan_device == pd) ? pd : -2;
    }
    if (!cfg.probes.empty()) {
      w.probe_ok = w.plan_device == w.device ? 1 : 0;
      uint64_t* pv = nullptr;
      CK(hexl_amd_device_alloc((void**)&pv, n * sizeof(uint64_t), w.device));
      std::vector<uint64_t> host(n);
      for (size_t i = 0; i < plans.size(); ++i) {
        const uint64_t q = moduli[w.segments[i].prime % moduli.size()];
        const ProbeDigest* want = nullptr;
        for (const ProbeDigest& d : cfg.probes)
          if (d.q == q) want = &d;
        if (!want) {
          w.probe_ok = 0;
          continue;
        }
        for (int inverse = 0; inverse < 2; ++inverse) {
          CK(hexl_amd_fill_splitmix(pv, n, 1, inverse ? 1001 : 1, q, stream));
          CK((inverse ? hexl_amd_ntt_inverse : hexl_amd_ntt_forward)(plans[i], pv, pv, 1, 1, 1, stream));
          CK(hexl_amd_copy(host.data(), pv, n * sizeof(uint64_t), stream, 1));
          if (Sha256::hex(host.data(), n * sizeof(uint64_t)) != (inverse ? want->inv : want->fwd)) w.probe_ok = 0;
        }
      }
      CK(hexl_amd_device_free(pv));
    }
    bool whole = w.segments.size() > 1;
    for (const Segment& s : w.segments) whole = whole && s.count == w.segments[0].count && s.first == 0;
    auto pass = [&](bool forward) -> int {
      if (whole) {  // several whole primes: prime-major blocks through the RNS entry point
        return (forward ? hexl_amd_ntt_forward_rns : hexl_amd_ntt_inverse_rns)(
            (const hexl_amd_ntt* const*)plans.data(), plans.size(), data, data, w.segments[0].count, 1, 1, stream);
      }
      for (size_t i = 0; i < w.segments.size(); ++i) {
        const int rc = (forward ? hexl_amd_ntt_forward : hexl_amd_ntt_inverse)(
            plans[i], seg_ptr[i], seg_ptr[i], w.segments[i].count, 1, 1, stream);
        if (rc) return rc;
      }
      return HEXL_AMD_OK;
    };
    auto fetch = [&](std::vector<uint64_t>& first, std::vector<uint64_t>& last) -> int {
      first.resize(n);
      last.resize(n);
      int rc = hexl_amd_copy(first.data(), data, n * sizeof(uint64_t), stream, 0);
      if (!rc) rc = hexl_amd_copy(last.data(), data + (w.polys - 1) * n, n * sizeof(uint64_t), stream, 1);
      return rc;
    };
    // verification pass (also the first warm-up step)
    CK(pass(true));
    CK(fetch(w.fwd_first, w.fwd_last));
    CK(pass(false));
    CK(fetch(w.back_first, w.back_last));
    for (int i = 0; i < cfg.warmup; ++i) {
      CK(pass(true));
      CK(pass(false));
    }
    CK(hexl_amd_synchronize(stream));
    bar.wait();  // ---- timed region starts (main reads the clock between the barriers too)
    passed_first = true;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < cfg.steps; ++i) {
      CK(pass(true));
      CK(pass(false));
    }
    CK(hexl_amd_synchronize(stream));
    w.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  body();
  if (!passed_first) bar.wait();  // (a worker that failed early still meets the others there)
  bar.wait();  // ---- timed region ends
  for (hexl_amd_ntt* p : plans) hexl_amd_ntt_destroy(p);
  if (data) hexl_amd_device_free(data);
  if (stream) hexl_amd_stream_destroy(stream);
}
#undef CK

// the same polynomial transformed alone on `device`: forward result and the generator's input
bool reference_poly(int device, uint64_t n, uint64_t q, uint64_t seed, std::vector<uint64_t>& input,
                    std::vector<uint64_t>& fwd, std::string& err) {
  hexl_amd_ntt* plan = nullptr;
  void* stream = nullptr;
  uint64_t* d = nullptr;
  input.resize(n);
  fwd.resize(n);
  int rc = hexl_amd_stream_create(&stream, device);
  if (!rc) rc = hexl_amd_ntt_create(&plan, n, q, 0, device);
  if (!rc) rc = hexl_amd_device_alloc((void**)&d, n * sizeof(uint64_t), device);
  if (!rc) rc = hexl_amd_fill_splitmix(d, n, 1, seed, q, stream);
  if (!rc) rc = hexl_amd_copy(input.data(), d, n * sizeof(uint64_t), stream, 1);
  if (!rc) rc = hexl_amd_ntt_forward(plan, d, d, 1, 1, 1, stream);
  if (!rc) rc = hexl_amd_copy(fwd.data(), d, n * sizeof(uint64_t), stream, 1);
  if (rc) err = hexl_amd_last_error();
  hexl_amd_ntt_destroy(plan);
  hexl_amd_device_free(d);
  hexl_amd_stream_destroy(stream);
  return rc == 0;
}

}  // namespace

int main(int argc, char** argv) {
  Config cfg;
  std::string devices_arg = "all";
  bool print_partition = false;  // the shards of the job as JSON, no device touched
  for (int i = 1; i < argc; ++i) {
    auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (!strcmp(argv[i], "--devices")) devices_arg = val();
    else if (!strcmp(argv[i], "--scaling")) cfg.weak = !strcmp(val(), "weak");
    else if (!strcmp(argv[i], "--n")) cfg.n = strtoull(val(), nullptr, 10);
    else if (!strcmp(argv[i], "--batch")) cfg.batch = strtoull(val(), nullptr, 10);
    else if (!strcmp(argv[i], "--primes")) cfg.primes = strtoull(val(), nullptr, 10);
    else if (!strcmp(argv[i], "--bits")) cfg.bits = strtoull(val(), nullptr, 10);
    else if (!strcmp(argv[i], "--steps")) cfg.steps = atoi(val());
    else if (!strcmp(argv[i], "--warmup")) cfg.warmup = atoi(val());
    else if (!strcmp(argv[i], "--ref-device")) cfg.ref_device = atoi(val());
    else if (!strcmp(argv[i], "--probe")) {
      const std::string a = val();
      const size_t c1 = a.find(':'), c2 = a.find(':', c1 == std::string::npos ? 0 : c1 + 1);
      if (c1 == std::string::npos || c2 == std::string::npos) return 2;
      cfg.probes.push_back({strtoull(a.substr(0, c1).c_str(), nullptr, 10), a.substr(c1 + 1, c2 - c1 - 1),
                            a.substr(c2 + 1)});
    } else if (!strcmp(argv[i], "--print-partition")) print_partition = true;
    else {
      fprintf(stderr, "unknown argument %s\n", argv[i]);
      return 2;
    }
  }
  int visible = 0;
  if (print_partition) {
    visible = 1 << 20;  // (any explicit device list is accepted)
    if (devices_arg == "all") return 2;
  } else if (hexl_amd_device_count(&visible) != HEXL_AMD_OK) {
    fprintf(stderr, "multi_device: %s\n", hexl_amd_last_error());
    return 3;
  }
  if (devices_arg == "all") {
    for (int d = 0; d < visible; ++d) cfg.devices.push_back(d);
  } else {
    for (size_t pos = 0; pos < devices_arg.size();) {
      size_t comma = devices_arg.find(',', pos);
      if (comma == std::string::npos) comma = devices_arg.size();
      const int d = atoi(devices_arg.substr(pos, comma - pos).c_str());
      if (d < 0 || d >= visible) {
        fprintf(stderr, "multi_device: device %d of %d\n", d, visible);
        return 2;
      }
      cfg.devices.push_back(d);
      pos = comma + 1;
    }
  }
  const uint64_t G = cfg.devices.size();
  if (G == 0 || cfg.steps < 1 || cfg.batch == 0 || cfg.primes == 0) return 2;
  if (cfg.ref_device < 0) cfg.ref_device = cfg.devices[0];

  if (print_partition) {
    const uint64_t total = (cfg.weak ? G : cfg.primes) * cfg.batch;
    printf("[");
    for (uint64_t g = 0; g < G; ++g) {
      const std::vector<Segment> segs = cfg.weak ? std::vector<Segment>{{g % cfg.primes, 0, cfg.batch}}
                                                 : shard_segments(total, cfg.batch, G, g);
      printf("%s[", g ? ", " : "");
      for (size_t i = 0; i < segs.size(); ++i)
        printf("%s[%llu, %llu, %llu]", i ? ", " : "", (unsigned long long)segs[i].prime,
               (unsigned long long)segs[i].first, (unsigned long long)segs[i].count);
      printf("]");
    }
    printf("]\n");
    return 0;
  }

  // GeneratePrimes(primes, bits, prefer_small, N): for 54 bits and N = 65536 the 8 RNS primes
  // of BASELINE configs[3]
  const uint64_t num_primes = cfg.weak ? std::max<uint64_t>(cfg.primes, 1) : cfg.primes;
  std::vector<uint64_t> moduli(num_primes);
  if (hexl_amd_generate_primes(moduli.data(), num_primes, cfg.bits, 1, cfg.n) != num_primes) {
    fprintf(stderr, "multi_device: not enough %llu-bit primes for N = %llu\n",
            (unsigned long long)cfg.bits, (unsigned long long)cfg.n);
    return 2;
  }

  std::vector<Worker> workers(G);
  const uint64_t job_primes = cfg.weak ? G : cfg.primes;
  const uint64_t total = job_primes * cfg.batch;
  for (uint64_t g = 0; g < G; ++g) {
    workers[g].index = (int)g;
    workers[g].device = cfg.devices[g];
    workers[g].segments = cfg.weak ? std::vector<Segment>{{g % num_primes, 0, cfg.batch}}
                                   : shard_segments(total, cfg.batch, G, g);
    for (const Segment& s : workers[g].segments) workers[g].polys += s.count;
    if (workers[g].polys == 0) {
      fprintf(stderr, "multi_device: more devices than units\n");
      return 2;
    }
  }

  Barrier bar((int)G + 1);
  std::vector<std::thread> threads;
  for (uint64_t g = 0; g < G; ++g)
    threads.emplace_back(run_worker, std::ref(workers[g]), std::cref(cfg), std::cref(moduli), std::ref(bar));
  bar.wait();
  const auto t0 = std::chrono::steady_clock::now();
  bar.wait();
  const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (auto& t : threads) t.join();

  bool ok = true;
  std::string error;
  for (const Worker& w : workers)
    if (!w.error.empty()) {
      ok = false;
      error = "worker " + std::to_string(w.index) + " (device " + std::to_string(w.device) + "): " + w.error;
      break;
    }
  if (ok)
    for (const Worker& w : workers)
      if (w.probe_ok == 0) {
        ok = false;
        error = "worker " + std::to_string(w.index) + " (device " + std::to_string(w.device) +
                "): per-rank probe failed (plan device " + std::to_string(w.plan_device) + ")";
        break;
      }
  // bit-compare every shard's first and last polynomial with a single-device run
  uint64_t compared = 0, mismatches = 0;
  if (ok) {
    for (const Worker& w : workers) {
      const Segment &sf = w.segments.front(), &sl = w.segments.back();
      struct Probe {
        uint64_t prime, poly;
        const std::vector<uint64_t>*fwd, *back;
      } probes[2] = {{sf.prime, sf.first, &w.fwd_first, &w.back_first},
                     {sl.prime, sl.first + sl.count - 1, &w.fwd_last, &w.back_last}};
      for (const Probe& p : probes) {
        std::vector<uint64_t> input, fwd;
        if (!reference_poly(cfg.ref_device, cfg.n, moduli[p.prime % moduli.size()],
                            1 + p.prime * cfg.batch + p.poly, input, fwd, error)) {
          ok = false;
          break;
        }
        compared += 2;
        if (fwd != *p.fwd) ++mismatches;
        if (input != *p.back) ++mismatches;
      }
      if (!ok) break;
    }
    if (mismatches) {
      ok = false;
      error = std::to_string(mismatches) + " of " + std::to_string(compared) +
              " probe polynomials differ from the single-device run";
    }
  }

  for (char& c : error)
    if (c == '"' || c == '\\' || c == '\n') c = ' ';
  const double ntts = 2.0 * (double)total * cfg.steps;
  printf("{\"launcher\": \"threads\", \"ok\": %s, \"error\": \"%s\", \"n_gpus\": %llu, \"devices\": [",
         ok ? "true" : "false", error.c_str(), (unsigned long long)G);
  for (uint64_t g = 0; g < G; ++g) printf("%s%d", g ? ", " : "", cfg.devices[g]);
  printf("], \"visible_devices\": %d, \"scaling\": \"%s\", \"N\": %llu, \"batch\": %llu, \"primes\": %llu, "
         "\"polynomials_total\": %llu, \"steps\": %d, \"warmup\": %d, \"value\": %.6g, \"unit\": \"NTT/s\", "
         "\"ms_per_step\": %.6g, \"per_rank_NTT_per_s\": [",
         visible, cfg.weak ? "weak" : "strong", (unsigned long long)cfg.n, (unsigned long long)cfg.batch,
         (unsigned long long)job_primes, (unsigned long long)total, cfg.steps, cfg.warmup,
         ok ? ntts / elapsed : 0.0, elapsed / cfg.steps * 1e3);
  for (uint64_t g = 0; g < G; ++g)
    printf("%s%.6g", g ? ", " : "",
           workers[g].seconds > 0 ? 2.0 * (double)workers[g].polys * cfg.steps / workers[g].seconds : 0.0);
  printf("], \"per_rank_polynomials\": [");
  for (uint64_t g = 0; g < G; ++g) printf("%s%llu", g ? ", " : "", (unsigned long long)workers[g].polys);
  printf("], \"per_rank_probe_ok\": [");
  for (uint64_t g = 0; g < G; ++g)
    printf("%s%s", g ? ", " : "", workers[g].probe_ok == 1 ? "true" : workers[g].probe_ok == 0 ? "false" : "null");
  printf("], \"per_rank_plan_device\": [");
  for (uint64_t g = 0; g < G; ++g) printf("%s%d", g ? ", " : "", workers[g].plan_device);
  printf("], \"probe_polynomials_compared\": %llu, \"probe_mismatches\": %llu, \"ref_device\": %d}\n",
         (unsigned long long)compared, (unsigned long long)mismatches, cfg.ref_device);
  return ok ? 0 : 1;
}
