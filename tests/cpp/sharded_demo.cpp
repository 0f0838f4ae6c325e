// A C++17 host sharding one logical filter over R contexts through the C ABI (include/beluga_mcl.h, "Particle shards"):
// one thread per rank, every rank with its own context and shard, a shared-memory transport between the threads
// (mcl_comm_attach) — what a host process driving the GPUs of one node does, here with all ranks on the GPU at hand so that
// the exchange logic runs on a one-GPU box.  On a node with several GPUs the only change is device_id = rank and
// mcl_comm_attach_rccl (or a peer-to-peer transport) in place of the host-staged one.
// Prints the estimates of the sharded filter and of a single-context filter on the same inputs; tests/test_cpp_facade.py
// compares them.  usage: sharded_demo [ranks = 2] [particles = 60000] [cycles = 6] [min_particles = particles] [estimate_kind = 0]
// estimate_kind 1: every update returns beluga::cluster_based_estimate (what beluga_ros::Amcl returns), over the shards as well.
// With min_particles < particles the filter is KLD-adaptive: the number of particles changes from cycle to cycle, every rank
// reports the same count, and it equals the single-context filter's.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "beluga_amd/amcl.hpp"
#include "beluga_mcl.h"

namespace {

struct Barrier {  // reusable; C++17 has no std::barrier
  std::mutex m;
  std::condition_variable cv;
  int count{0}, generation{0}, parties;
  explicit Barrier(int n) : parties(n) {}
  void wait() {
    std::unique_lock<std::mutex> lock(m);
    const int gen = generation;
    if (++count == parties) {
      count = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return gen != generation; });
    }
  }
};

// Host-staged exchange between the threads of this process.
struct Exchange {
  int world;
  Barrier barrier;
  std::vector<std::vector<char>> boxes;                // all_gather: one box per rank
  std::vector<std::vector<std::vector<char>>> mail;    // all_to_all: mail[from][to]
  explicit Exchange(int n) : world(n), barrier(n), boxes(n), mail(n, std::vector<std::vector<char>>(n)) {}
};
struct Endpoint {
  Exchange* x;
  int rank;
};

int32_t all_gather(void* user, const void* d_send, void* d_recv, uint64_t bytes, void* stream) {
  auto* e = static_cast<Endpoint*>(user);
  if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return 1;
  e->x->boxes[e->rank].resize(bytes);
  if (hipMemcpy(e->x->boxes[e->rank].data(), d_send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  e->x->barrier.wait();
  for (int r = 0; r < e->x->world; ++r)
    if (hipMemcpy(static_cast<char*>(d_recv) + r * bytes, e->x->boxes[r].data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
  e->x->barrier.wait();
  return 0;
}
int32_t all_to_all(void* user, const void* d_send, const uint64_t* send_bytes, void* d_recv, const uint64_t* recv_bytes, void* stream) {
  auto* e = static_cast<Endpoint*>(user);
  if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return 1;
  const char* out = static_cast<const char*>(d_send);
  for (int q = 0; q < e->x->world; ++q) {
    auto& box = e->x->mail[e->rank][q];
    box.resize(send_bytes[q]);
    if (send_bytes[q] && hipMemcpy(box.data(), out, send_bytes[q], hipMemcpyDeviceToHost) != hipSuccess) return 1;
    out += send_bytes[q];
  }
  e->x->barrier.wait();
  char* in = static_cast<char*>(d_recv);
  for (int q = 0; q < e->x->world; ++q) {
    const auto& box = e->x->mail[q][e->rank];
    if (box.size() != recv_bytes[q]) return 2;
    if (recv_bytes[q] && hipMemcpy(in, box.data(), recv_bytes[q], hipMemcpyHostToDevice) != hipSuccess) return 1;
    in += recv_bytes[q];
  }
  e->x->barrier.wait();
  return 0;
}

struct Scenario {
  uint32_t W = 200, H = 160;
  std::vector<int8_t> cells;
  std::vector<std::vector<double>> scans;
  std::vector<std::vector<double>> controls;
  Scenario(int cycles) : cells(W * H, 0) {
    for (uint32_t x = 0; x < W; ++x) cells[120 * W + x] = cells[10 * W + x] = 100;
    for (uint32_t y = 0; y < H; ++y) cells[y * W + 150] = cells[y * W + 5] = 100;
    double ox = 0, oy = 0, ot = 0;
    for (int c = 0; c < cycles; ++c) {
      ox += 0.3 * std::cos(ot);
      oy += 0.3 * std::sin(ot);
      ot += 0.04;
      controls.push_back({std::cos(ot), std::sin(ot), ox, oy});
      std::vector<double> scan;
      for (int b = 0; b < 120; ++b) {
        const double a = -2.0 + b * (4.0 / 120), r = 2.0 + 0.5 * std::sin(0.3 * b + c);
        scan.push_back(r * std::cos(a));
        scan.push_back(r * std::sin(a));
      }
      scans.push_back(scan);
    }
  }
};

uint64_t g_min_particles = 0;  // 0: fixed size
int g_estimate_kind = 0;       // 1: cluster_based_estimate

mcl_config make_config(uint64_t n_total, uint64_t shard_offset, uint64_t shard_capacity) {
  mcl_config cfg;
  mcl_default_config(&cfg);
  cfg.seed = 77;
  cfg.amcl.min_particles = cfg.amcl.max_particles = n_total;
  if (g_min_particles) {
    cfg.amcl.min_particles = g_min_particles;
    cfg.amcl.kld_epsilon = 0.05;
    cfg.amcl.kld_z = 3.0;
    cfg.amcl.spatial_resolution_x = cfg.amcl.spatial_resolution_y = 0.2;
    cfg.amcl.spatial_resolution_theta = 0.1;
  }
  cfg.motion = mcl_diffdrive_params{0.1, 0.05, 0.1, 0.05, 0.01};
  cfg.lf = mcl_lf_params{2.0, 100.0, 0.5, 0.5, 0.2, 1, 0};
  cfg.shard_offset = shard_offset;
  cfg.shard_capacity = shard_capacity;
  return cfg;
}

bool run_filter(const Scenario& sc, mcl_ctx* ctx, std::vector<mcl_estimate>* out, std::vector<uint64_t>* counts = nullptr) {
  const double origin[4] = {1.0, 0.0, -2.0, -3.0};
  const int8_t traits[3] = {0, -1, 100};
  if (mcl_set_map(ctx, sc.cells.data(), sc.W, sc.H, 0.05, origin, traits) != MCL_OK) return false;
  if (g_estimate_kind && mcl_set_estimate_kind(ctx, g_estimate_kind, nullptr) != MCL_OK) return false;
  const double mean[3] = {1.0, 1.0, 0.2}, cov[9] = {0.09, 0, 0, 0, 0.09, 0, 0, 0, 0.02};
  if (mcl_initialize_normal(ctx, mean, cov) != MCL_OK) return false;
  for (size_t c = 0; c < sc.scans.size(); ++c) {
    if (c == 3) {
      // the caller takes the set out and puts it back (what a host does to checkpoint / restore a filter): a sharded context
      // then has to find out the size of the whole set and where its own shard starts from the other ranks
      uint64_t held = 0, got = 0;
      if (mcl_num_particles(ctx, &held) != MCL_OK) return false;
      std::vector<double> states(4 * held), weights(held);
      if (mcl_get_particles(ctx, states.data(), weights.data(), held, &got) != MCL_OK || got != held) return false;
      if (mcl_set_particles(ctx, states.data(), weights.data(), held) != MCL_OK) return false;
    }
    mcl_estimate est;
    mcl_update_info info;
    if (mcl_update(ctx, sc.controls[c].data(), sc.scans[c].data(), sc.scans[c].size() / 2, &est, &info) != MCL_OK) {
      std::printf("update_error %s\n", mcl_last_error(ctx));
      return false;
    }
    if (!info.updated || !info.resampled) return false;
    out->push_back(est);
    if (counts) counts->push_back(info.num_particles);
  }
  return true;
}

// The same sharded filter through the C++ facade (include/beluga_amd/amcl.hpp): one beluga_amd::Amcl per rank, constructed with
// its Shard, attached to the exchange; every rank's estimates, cycle by cycle.
bool run_facade_shards(const Scenario& sc, int ranks, uint64_t n_total, std::vector<std::vector<mcl_estimate>>* out) {
  using namespace beluga_amd;
  Exchange exchange(ranks);
  std::vector<Endpoint> endpoints(ranks);
  std::vector<int> failed(ranks, 0);
  std::vector<std::thread> threads;
  out->assign(ranks, {});
  for (int r = 0; r < ranks; ++r) {
    endpoints[r] = Endpoint{&exchange, r};
    threads.emplace_back([&, r] {
      try {
        OccupancyGridView map;
        map.cells = sc.cells.data();
        map.width = sc.W;
        map.height = sc.H;
        map.resolution = 0.05;
        map.origin = SE2d{0.0, -2.0, -3.0};
        AmclParams params;
        params.min_particles = params.max_particles = n_total;
        if (g_min_particles) {
          params.min_particles = g_min_particles;
          params.kld_epsilon = 0.05;
          params.kld_z = 3.0;
          params.spatial_resolution_x = params.spatial_resolution_y = 0.2;
          params.spatial_resolution_theta = 0.1;
        }
        LikelihoodFieldModelParam lf;
        lf.max_obstacle_distance = 2.0;
        lf.max_laser_distance = 100.0;
        lf.model_unknown_space = true;
        Amcl filter{map, DifferentialDriveModelParam{0.1, 0.05, 0.1, 0.05}, lf, params, /*seed=*/77, /*device=*/0, {},
                    Shard::of(n_total, static_cast<unsigned>(r), static_cast<unsigned>(ranks))};
        filter.attach(static_cast<unsigned>(r), static_cast<unsigned>(ranks), mcl_transport{&endpoints[r], all_gather, all_to_all});
        if (g_estimate_kind) filter.use_cluster_based_estimate(true);
        filter.initialize(SE2d{0.2, 1.0, 1.0}, Matrix3d{0.09, 0, 0, 0, 0.09, 0, 0, 0, 0.02});
        for (size_t c = 0; c < sc.scans.size(); ++c) {
          Amcl::measurement_type scan;
          for (size_t b = 0; b + 1 < sc.scans[c].size(); b += 2) scan.emplace_back(sc.scans[c][b], sc.scans[c][b + 1]);
          SE2d control;
          std::memcpy(control.data(), sc.controls[c].data(), 4 * sizeof(double));
          const auto est = filter.update(control, scan);
          if (!est) {
            failed[r] = 1;
            return;
          }
          mcl_estimate e{};
          std::memcpy(e.pose, est->first.data(), 4 * sizeof(double));
          std::memcpy(e.covariance, est->second.data(), 9 * sizeof(double));
          (*out)[r].push_back(e);
        }
      } catch (const std::exception& e) {
        std::printf("facade_error %s\n", e.what());
        failed[r] = 1;
      }
    });
  }
  for (auto& t : threads) t.join();
  for (int r = 0; r < ranks; ++r)
    if (failed[r]) return false;
  return true;
}

}  // namespace

int main(int argc, char** argv) {
  const int ranks = argc > 1 ? std::atoi(argv[1]) : 2;
  const uint64_t n_total = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 60000;
  const int cycles = argc > 3 ? std::atoi(argv[3]) : 6;
  g_min_particles = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 0;
  if (g_min_particles >= n_total) g_min_particles = 0;
  g_estimate_kind = argc > 5 ? std::atoi(argv[5]) : 0;
  const bool mismatch = argc > 6 && std::string(argv[6]) == "mismatch";
  const Scenario sc(cycles);
  if (mismatch) {
    // One rank runs another configuration (what a stray BELUGA_MCL_DEVICE_POLICY in one process's environment does): the
    // communicator's first collective compares the ranks' configurations, and EVERY rank's attach fails - none is left waiting
    // in a collective the others never enter.
    Exchange exchange(ranks);
    std::vector<Endpoint> endpoints(ranks);
    std::vector<int> refused(ranks, 0);
    std::vector<std::thread> threads;
    for (int r = 0; r < ranks; ++r) {
      endpoints[r] = Endpoint{&exchange, r};
      threads.emplace_back([&, r] {
        const uint64_t base = n_total / ranks, rem = n_total % ranks;
        const uint64_t first = r * base + std::min<uint64_t>(r, rem), mine = base + (static_cast<uint64_t>(r) < rem ? 1 : 0);
        mcl_ctx* ctx = nullptr;
        const mcl_config cfg = make_config(n_total, first, mine);
        if (mcl_create(&cfg, &ctx) != MCL_OK) return;
        if (r == 1) mcl_set_option(ctx, "device_policy", 0);
        const mcl_transport transport{&endpoints[r], all_gather, all_to_all};
        const mcl_status st = mcl_comm_attach(ctx, static_cast<uint32_t>(r), static_cast<uint32_t>(ranks), &transport);
        refused[r] = st == MCL_ERR_INVALID_ARGUMENT && std::string(mcl_last_error(ctx)).find("another configuration") != std::string::npos;
        mcl_destroy(ctx);
      });
    }
    for (auto& t : threads) t.join();
    int total = 0;
    for (int r = 0; r < ranks; ++r) total += refused[r];
    std::printf("attach_refused %d of %d\n", total, ranks);
    return total == ranks ? 0 : 8;
  }

  mcl_ctx* single = nullptr;
  const mcl_config whole = make_config(n_total, 0, 0);
  if (mcl_create(&whole, &single) != MCL_OK) {
    std::printf("runtime_error %s\n", mcl_last_error(nullptr));
    return 3;
  }
  std::vector<mcl_estimate> reference;
  std::vector<uint64_t> reference_counts;
  if (!run_filter(sc, single, &reference, &reference_counts)) return 4;
  std::vector<double> ref_states(4 * n_total), ref_weights(n_total);
  uint64_t got = 0;
  mcl_get_particles(single, ref_states.data(), ref_weights.data(), n_total, &got);
  mcl_destroy(single);

  Exchange exchange(ranks);
  std::vector<Endpoint> endpoints(ranks);
  std::vector<std::vector<mcl_estimate>> estimates(ranks);
  std::vector<std::vector<double>> shard_states(ranks);
  std::vector<std::vector<uint64_t>> counts(ranks);
  std::vector<int> status(ranks, 0);
  std::vector<std::thread> threads;
  for (int r = 0; r < ranks; ++r) {
    endpoints[r] = Endpoint{&exchange, r};
    threads.emplace_back([&, r] {
      const uint64_t base = n_total / ranks, rem = n_total % ranks;
      const uint64_t first = r * base + std::min<uint64_t>(r, rem), mine = base + (static_cast<uint64_t>(r) < rem ? 1 : 0);
      mcl_ctx* ctx = nullptr;
      const mcl_config cfg = make_config(n_total, first, mine);
      if (mcl_create(&cfg, &ctx) != MCL_OK) {
        status[r] = 1;
        return;
      }
      const mcl_transport transport{&endpoints[r], all_gather, all_to_all};
      if (mcl_comm_attach(ctx, static_cast<uint32_t>(r), static_cast<uint32_t>(ranks), &transport) != MCL_OK) status[r] = 2;
      if (!status[r] && !run_filter(sc, ctx, &estimates[r], &counts[r])) status[r] = 3;
      if (!status[r]) {
        uint64_t held = 0;
        if (mcl_num_particles(ctx, &held) != MCL_OK || held > mine || (!g_min_particles && held != mine)) status[r] = 4;
        shard_states[r].resize(4 * held);
        std::vector<double> w(held);
        uint64_t n = 0;
        if (!status[r] && (mcl_get_particles(ctx, shard_states[r].data(), w.data(), held, &n) != MCL_OK || n != held)) status[r] = 4;
      }
      mcl_destroy(ctx);
    });
  }
  for (auto& t : threads) t.join();
  for (int r = 0; r < ranks; ++r)
    if (status[r]) {
      std::printf("rank_failed %d %d\n", r, status[r]);
      return 5;
    }
  double worst_pose = 0, worst_cov = 0;
  for (int c = 0; c < cycles; ++c) {
    for (int r = 0; r < ranks; ++r) {  // every rank returns the same estimate
      if (std::memcmp(&estimates[r][c], &estimates[0][c], sizeof(mcl_estimate)) != 0) {
        std::printf("ranks_disagree %d %d\n", c, r);
        return 6;
      }
    }
    for (int k = 0; k < 4; ++k) worst_pose = std::max(worst_pose, std::abs(estimates[0][c].pose[k] - reference[c].pose[k]));
    for (int k = 0; k < 9; ++k) worst_cov = std::max(worst_cov, std::abs(estimates[0][c].covariance[k] - reference[c].covariance[k]));
  }
  // the particle counts (KLD-adaptive filters: an integer result, the same on every rank and in the single-context filter)
  uint64_t count_mismatches = 0;
  for (int c = 0; c < cycles; ++c)
    for (int r = 0; r < ranks; ++r)
      if (counts[r][c] != reference_counts[c]) ++count_mismatches;
  uint64_t total_held = 0;
  for (int r = 0; r < ranks; ++r) total_held += shard_states[r].size() / 4;
  std::printf("particle_counts");
  for (int c = 0; c < cycles; ++c) std::printf(" %llu", static_cast<unsigned long long>(reference_counts[c]));
  std::printf("\ncount_mismatches %llu\nparticles_held %llu %llu\n", static_cast<unsigned long long>(count_mismatches),
              static_cast<unsigned long long>(total_held), static_cast<unsigned long long>(got));
  // the sharded set, concatenated in rank order, against the single-context set
  uint64_t different = 0, at = 0;
  for (int r = 0; r < ranks; ++r)
    for (size_t i = 0; i < shard_states[r].size() / 4; ++i, ++at)
      if (std::memcmp(&shard_states[r][4 * i], &ref_states[4 * at], 4 * sizeof(double)) != 0) ++different;
  // the facade's shards against the C ABI's: the same library calls underneath, so the same bits
  std::vector<std::vector<mcl_estimate>> facade;
  if (!run_facade_shards(sc, ranks, n_total, &facade)) return 7;
  uint64_t facade_mismatches = 0;
  for (int c = 0; c < cycles; ++c)
    for (int r = 0; r < ranks; ++r)
      if (std::memcmp(facade[r][c].pose, estimates[0][c].pose, sizeof(estimates[0][c].pose)) != 0 ||
          std::memcmp(facade[r][c].covariance, estimates[0][c].covariance, sizeof(estimates[0][c].covariance)) != 0)
        ++facade_mismatches;
  std::printf("facade_mismatches %llu\n", static_cast<unsigned long long>(facade_mismatches));
  std::printf("ranks %d particles %llu cycles %d\n", ranks, static_cast<unsigned long long>(n_total), cycles);
  std::printf("estimate_max_abs_difference %.3e %.3e\n", worst_pose, worst_cov);
  std::printf("particles_that_differ %llu\n", static_cast<unsigned long long>(different));
  return 0;
}
