// One RANK of a sharded filter as a process of its own (tests/test_cpp_facade.py starts `world` of them, and one with world = 1
// as the single-context reference): what a host that runs one process per GPU does - mcl_create with this rank's shard,
// mcl_comm_attach with a transport, mcl_update every cycle - with all ranks on the GPU at hand and a host-staged transport through
// POSIX shared memory (a process-shared barrier, one gather box per rank, one mail box per pair of ranks), so that the
// library's sharded cycle (sharded_update: the collectives, their sizes, the host synchronisations) runs between real processes on
// a one-GPU box.  On a node with several GPUs: device_id = rank and mcl_comm_attach_rccl instead.
//
//   sharded_procs <shm name> <rank> <world> <particles> <cycles> <out file> [shard_pad_permille = library default (-1)]
//                 [alpha_slow alpha_fast = library defaults; "recovery" scenario: before the middle cycle the two filters of the recovery
//                  estimator are put apart (mcl_debug_set_recovery_filters: slow = 2 / N, fast = 1 / N - with a fixed particle count the
//                  estimator never leaves p = 0 by itself), and that cycle resamples with a random state probability of several tenths]
//
// Prints, per cycle: the estimate (13 doubles, hex floats) and what the cycle added to the library's communication counters
// (collectives, bytes handed to the transport, host synchronisations); at the end the exchange's overflow count.  Writes the
// rank's particle states (4 doubles each) to <out file>.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "beluga_mcl.h"

namespace {

struct Header {
  std::atomic<uint32_t> ready;  // set by rank 0 once the barrier is initialised
  uint32_t world;
  uint64_t box_bytes, mail_bytes;
  pthread_barrier_t barrier;
  uint64_t mail_size[64 * 64];  // bytes in mail[from][to]
};
struct Shared {
  Header* h{nullptr};
  char* boxes{nullptr};  // [world][box_bytes]
  char* mail{nullptr};   // [world][world][mail_bytes]
  int rank{0};
};

int32_t all_gather(void* user, const void* d_send, void* d_recv, uint64_t bytes, void* stream) {
  auto* s = static_cast<Shared*>(user);
  if (bytes > s->h->box_bytes) return 3;
  if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return 1;
  if (hipMemcpy(s->boxes + s->rank * s->h->box_bytes, d_send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  pthread_barrier_wait(&s->h->barrier);
  for (uint32_t r = 0; r < s->h->world; ++r)
    if (hipMemcpy(static_cast<char*>(d_recv) + r * bytes, s->boxes + r * s->h->box_bytes, bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
  pthread_barrier_wait(&s->h->barrier);
  return 0;
}
int32_t all_to_all(void* user, const void* d_send, const uint64_t* send_bytes, void* d_recv, const uint64_t* recv_bytes, void* stream) {
  auto* s = static_cast<Shared*>(user);
  const uint32_t world = s->h->world;
  if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return 1;
  const char* out = static_cast<const char*>(d_send);
  for (uint32_t q = 0; q < world; ++q) {
    if (send_bytes[q] > s->h->mail_bytes) return 3;
    char* box = s->mail + (static_cast<uint64_t>(s->rank) * world + q) * s->h->mail_bytes;
    s->h->mail_size[s->rank * 64 + q] = send_bytes[q];
    if (send_bytes[q] && hipMemcpy(box, out, send_bytes[q], hipMemcpyDeviceToHost) != hipSuccess) return 1;
    out += send_bytes[q];
  }
  pthread_barrier_wait(&s->h->barrier);
  char* in = static_cast<char*>(d_recv);
  for (uint32_t q = 0; q < world; ++q) {
    const char* box = s->mail + (static_cast<uint64_t>(q) * world + s->rank) * s->h->mail_bytes;
    if (s->h->mail_size[q * 64 + s->rank] != recv_bytes[q]) return 2;  // the two sides disagree about a message's size
    if (recv_bytes[q] && hipMemcpy(in, box, recv_bytes[q], hipMemcpyHostToDevice) != hipSuccess) return 1;
    in += recv_bytes[q];
  }
  pthread_barrier_wait(&s->h->barrier);
  return 0;
}

uint64_t counter(mcl_ctx* ctx, const char* name) {
  uint64_t v = 0;
  mcl_get_counter(ctx, name, &v);
  return v;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 7) {
    std::printf("usage: sharded_procs <shm> <rank> <world> <particles> <cycles> <out> [pad_permille]\n");
    return 64;
  }
  const std::string shm_name = argv[1];
  const int rank = std::atoi(argv[2]), world = std::atoi(argv[3]);
  const uint64_t n_total = std::strtoull(argv[4], nullptr, 10);
  const int cycles = std::atoi(argv[5]);
  const std::string out_path = argv[6];
  const long pad = argc > 7 ? std::atol(argv[7]) : -1;
  const bool recovery = argc > 9;
  const double alpha_slow = recovery ? std::atof(argv[8]) : 0.0, alpha_fast = recovery ? std::atof(argv[9]) : 0.0;
  if (world < 1 || world > 64 || rank < 0 || rank >= world) return 64;

  // the scenario of tests/cpp/sharded_demo.cpp: a 200 x 160 map with four walls, 120-beam scans, a gentle arc
  const uint32_t W = 200, H = 160;
  std::vector<int8_t> cells(W * H, 0);
  for (uint32_t x = 0; x < W; ++x) cells[120 * W + x] = cells[10 * W + x] = 100;
  for (uint32_t y = 0; y < H; ++y) cells[y * W + 150] = cells[y * W + 5] = 100;

  Shared shared;
  shared.rank = rank;
  size_t shm_bytes = 0;
  if (world > 1) {
    // a mail box holds what one rank sends ONE peer: a shard's every request to it where that is small, 2.5 x the average share of a
    // shard's output slots (states of 32 bytes; the fixed capacity is 1.07 x) at the sizes where world^2 such boxes would not fit /dev/shm
    const uint64_t per_shard = n_total / world + 1;
    const uint64_t box_bytes = 1 << 16;
    const uint64_t mail_bytes = (std::min<uint64_t>(per_shard, std::max<uint64_t>(1u << 17, 5 * (per_shard / world) / 2)) * 32 + 4095) & ~4095ull;
    shm_bytes = sizeof(Header) + world * box_bytes + static_cast<uint64_t>(world) * world * mail_bytes;
    int fd = -1;
    if (rank == 0) {
      shm_unlink(shm_name.c_str());
      fd = shm_open(shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, static_cast<off_t>(shm_bytes)) != 0) return 65;
    } else {
      for (int tries = 0; tries < 20000 && fd < 0; ++tries) {  // (rank 0 creates it)
        fd = shm_open(shm_name.c_str(), O_RDWR, 0600);
        struct stat st;
        if (fd >= 0 && (fstat(fd, &st) != 0 || static_cast<size_t>(st.st_size) < shm_bytes)) {
          close(fd);
          fd = -1;
        }
        if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
      if (fd < 0) return 65;
    }
    void* base = mmap(nullptr, shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) return 65;
    shared.h = static_cast<Header*>(base);
    shared.boxes = static_cast<char*>(base) + sizeof(Header);
    shared.mail = shared.boxes + world * box_bytes;
    if (rank == 0) {
      shared.h->world = static_cast<uint32_t>(world);
      shared.h->box_bytes = box_bytes;
      shared.h->mail_bytes = mail_bytes;
      pthread_barrierattr_t attr;
      pthread_barrierattr_init(&attr);
      pthread_barrierattr_setpshared(&attr, PTHREAD_PROCESS_SHARED);
      pthread_barrier_init(&shared.h->barrier, &attr, static_cast<unsigned>(world));
      shared.h->ready.store(1u, std::memory_order_release);
    } else {
      for (int tries = 0; tries < 20000 && shared.h->ready.load(std::memory_order_acquire) == 0u; ++tries)
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      if (shared.h->ready.load(std::memory_order_acquire) == 0u) return 65;
    }
  }

  const uint64_t base = n_total / world, rem = n_total % world;
  const uint64_t first = rank * base + std::min<uint64_t>(rank, rem), mine = base + (static_cast<uint64_t>(rank) < rem ? 1 : 0);
  mcl_config cfg;
  mcl_default_config(&cfg);
  cfg.seed = 77;
  cfg.amcl.min_particles = cfg.amcl.max_particles = n_total;
  cfg.motion = mcl_diffdrive_params{0.1, 0.05, 0.1, 0.05, 0.01};
  cfg.lf = mcl_lf_params{2.0, 100.0, 0.5, 0.5, 0.2, 1, 0};
  if (recovery) {
    cfg.amcl.alpha_slow = alpha_slow;
    cfg.amcl.alpha_fast = alpha_fast;
  }
  if (world > 1) {
    cfg.shard_offset = first;
    cfg.shard_capacity = mine;
  }
  mcl_ctx* ctx = nullptr;
  if (mcl_create(&cfg, &ctx) != MCL_OK) {
    std::printf("runtime_error %s\n", mcl_last_error(nullptr));
    return 3;
  }
  if (pad >= 0 && mcl_set_option(ctx, "shard_pad_permille", pad) != MCL_OK) return 4;
  const mcl_transport transport{&shared, all_gather, all_to_all};
  if (world > 1 && mcl_comm_attach(ctx, static_cast<uint32_t>(rank), static_cast<uint32_t>(world), &transport) != MCL_OK) {
    std::printf("attach_error %s\n", mcl_last_error(ctx));
    return 5;
  }
  const double origin[4] = {1.0, 0.0, -2.0, -3.0};
  const int8_t traits[3] = {0, -1, 100};
  if (mcl_set_map(ctx, cells.data(), W, H, 0.05, origin, traits) != MCL_OK) return 6;
  const double mean[3] = {1.0, 1.0, 0.2}, cov[9] = {0.09, 0, 0, 0, 0.09, 0, 0, 0, 0.02};
  if (mcl_initialize_normal(ctx, mean, cov) != MCL_OK) return 6;
  double ox = 0, oy = 0, ot = 0;
  for (int c = 0; c < cycles; ++c) {
    ox += 0.3 * std::cos(ot);
    oy += 0.3 * std::sin(ot);
    ot += 0.04;
    const double control[4] = {std::cos(ot), std::sin(ot), ox, oy};
    std::vector<double> scan;
    for (int b = 0; b < 120; ++b) {
      const double a = -2.0 + b * (4.0 / 120), r = 2.0 + 0.5 * std::sin(0.3 * b + c);
      scan.push_back(r * std::cos(a));
      scan.push_back(r * std::sin(a));
    }
    if (recovery && c == cycles / 2 &&
        mcl_debug_set_recovery_filters(ctx, 2.0 / static_cast<double>(n_total), 1.0 / static_cast<double>(n_total)) != MCL_OK) return 7;
    const uint64_t collectives = counter(ctx, "comm_collectives"), bytes = counter(ctx, "comm_bytes_out"), syncs = counter(ctx, "comm_host_syncs");
    mcl_estimate est;
    mcl_update_info info;
    if (mcl_update(ctx, control, scan.data(), scan.size() / 2, &est, &info) != MCL_OK) {
      std::printf("update_error %s\n", mcl_last_error(ctx));
      return 7;
    }
    std::printf("cycle %d est", c);
    for (int k = 0; k < 4; ++k) std::printf(" %a", est.pose[k]);
    for (int k = 0; k < 9; ++k) std::printf(" %a", est.covariance[k]);
    std::printf(" collectives %llu bytes %llu syncs %llu resampled %d n %llu p_permille %llu\n",
                static_cast<unsigned long long>(counter(ctx, "comm_collectives") - collectives),
                static_cast<unsigned long long>(counter(ctx, "comm_bytes_out") - bytes),
                static_cast<unsigned long long>(counter(ctx, "comm_host_syncs") - syncs), info.resampled,
                static_cast<unsigned long long>(info.num_particles),
                static_cast<unsigned long long>(info.random_state_probability * 1000.0));
  }
  std::printf("overflows %llu\n", static_cast<unsigned long long>(counter(ctx, "comm_overflows")));
  uint64_t held = 0, got = 0;
  if (mcl_num_particles(ctx, &held) != MCL_OK) return 8;
  std::vector<double> states(4 * held), weights(held);
  if (mcl_get_particles(ctx, states.data(), weights.data(), held, &got) != MCL_OK || got != held) return 8;
  if (FILE* f = std::fopen(out_path.c_str(), "wb")) {
    std::fwrite(states.data(), sizeof(double), states.size(), f);
    std::fclose(f);
  } else {
    return 9;
  }
  std::printf("held %llu\n", static_cast<unsigned long long>(held));
  mcl_destroy(ctx);
  if (world > 1) {
    pthread_barrier_wait(&shared.h->barrier);  // nobody unmaps (or unlinks) while a peer is still inside a collective
    munmap(shared.h, shm_bytes);
    if (rank == 0) shm_unlink(shm_name.c_str());
  }
  return 0;
}
