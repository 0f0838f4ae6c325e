"""Particle-sharded MCL update across the GPUs of one node: one process per GPU, torch.distributed (RCCL) between them.

The reference is single-process (no collectives exist upstream).  The path shards naturally:
propagate / reweight are independent per particle; the couplings are
  C1  all-reduce of the weight sum (actions/normalize.hpp:70 becomes a global sum),
  C2  all-gather of {shard CDF total, sum w, sum w^2} (discrete_distribution's partial sums, ESS, Thrun average),
  C3  the ancestor exchange of multinomial resampling: every output slot j (global index space) draws
      u_j from the SAME Philox stream as the single-GPU path, finds the shard that owns that point of the global
      CDF, and fetches the ancestor's state from it (all-to-all of 8-byte targets out, 32-byte states back),
  C4  all-reduce of the nine estimate sums (algorithm/estimation.hpp:436-475),
  C5  (KLD-adaptive mode, min_particles < max_particles) the candidate stream of take_while_kld is drawn block by block,
      each rank drawing a slice of the block through C3; the 8-byte spatial hashes of the block are all-gathered so that
      every rank evaluates kld_condition over the same global sequence and finds the same cut; the kept candidates are
      then re-balanced into contiguous shards of the new (smaller) set with one all-to-all per block.
Because every random number is addressed by global index, the sharded filter reproduces the single-GPU
particle set for any number of ranks (up to the rounding of the summation order).

`ShardedAmcl` has the same surface as `beluga_amd.Amcl` (initialize / update / particles / force_update).
The per-rank compute goes through an *engine*; the product engine is `HipShardEngine` (the C ABI).  Tests
drive the same orchestration on CPU ranks (gloo) with an engine of their own.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from .amcl import AmclParams, estimate_from_sums


# ---- host SE2 helpers (policies only; per-particle math lives in the engine) --------------------------
def _se2_mul(a, b):
    c = a[0] * b[0] - a[1] * b[1]
    s = a[0] * b[1] + a[1] * b[0]
    n = math.hypot(c, s)
    return np.array([c / n, s / n, a[2] + a[0] * b[2] - a[1] * b[3], a[3] + a[1] * b[2] + a[0] * b[3]])


def _se2_inverse(a):
    c, s = a[0], -a[1]
    return np.array([c, s, c * -a[2] - s * -a[3], s * -a[2] + c * -a[3]])


class _ExponentialFilter:  # algorithm/exponential_filter.hpp:32-44
    def __init__(self, alpha):
        self.alpha, self.output = alpha, 0.0

    def reset(self):
        self.output = 0.0

    def __call__(self, x):
        self.output += x if self.output == 0.0 else self.alpha * (x - self.output)
        return self.output


class HipShardEngine:
    """One shard of the particle set on one GPU, through libbeluga_mcl.so. Launches go on torch's current stream."""

    def __init__(self, grid, motion, sensor, params: AmclParams, seed: int, device: int, shard_offset: int, shard_capacity: int):
        import torch

        from .amcl import Amcl
        self.torch = torch
        self.device = torch.device("cuda", device)
        # One explicit stream shared by the library's kernels, torch's glue ops and the RCCL collectives
        # (torch's default stream is the NULL stream, which the library would not adopt).
        self.stream = torch.cuda.Stream(self.device)
        self.f = Amcl(grid, motion, sensor, params, seed=seed, device=device, shard_offset=shard_offset,
                      shard_capacity=shard_capacity, hip_stream=self.stream.cuda_stream)

    def stream_scope(self):
        return self.torch.cuda.stream(self.stream)

    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.float64, device=self.device)

    def initialize(self, pose, cov):
        self.f.initialize(pose, cov)

    def set_particles(self, states, weights):
        self.f.set_particles(states, weights)

    def particles(self):
        return self.f.particles()

    def num_particles(self):
        return self.f.num_particles()

    def propagate(self, pose, prev, step):
        self.f.propagate(pose, prev, step)

    def reweight(self, points):
        self.f.reweight(points)

    # scalars stay on the device: tensors in, tensors out, nothing synchronises
    def weight_sum_into(self, t_sum):
        self.f.weight_sum_device(t_sum.data_ptr())

    def normalize_from(self, t_factor, t_stats2):
        self.f.normalize_device(t_factor.data_ptr(), t_stats2.data_ptr())

    def build_cdf_into(self, t_total):
        self.f.build_cdf_device(t_total.data_ptr())

    def estimate_sums_into(self, pivot, t_sums9):
        self.f.estimate_sums_device(pivot, t_sums9.data_ptr())

    def resample_targets(self, step, p, total, first_slot, count, targets):
        self.f.resample_targets(step, p, total, first_slot, count, targets.data_ptr())

    def route_targets(self, targets, ends, offsets, self_rank):
        """-> (shard-local targets grouped by owner rank, order[k] = slot answered by request k, int64 counts per rank)"""
        torch, m, world = self.torch, targets.numel(), ends.numel()
        send = self.empty(m)
        order = torch.empty(m, dtype=torch.int32, device=self.device)
        counts = torch.empty(world, dtype=torch.int64, device=self.device)
        self.f.route_targets(targets.data_ptr(), m, ends.data_ptr(), offsets.data_ptr(), world, self_rank, send.data_ptr(),
                             order.data_ptr(), counts.data_ptr())
        return send, order, counts

    def serve_requests(self, requests):
        replies = self.empty(requests.numel(), 4)
        self.f.serve_requests(requests.data_ptr(), requests.numel(), replies.data_ptr())
        return replies

    def commit_routed(self, step, first_slot, count, replies, order, targets):
        self.f.commit_routed(step, first_slot, count, replies.data_ptr(), order.data_ptr(), targets.data_ptr())

    def finish_candidates(self, step, first_slot, count, replies, order, targets):
        """-> (states (count, 4) as (cos, sin, x, y) in slot order, spatial hashes (count,) as int64 bit patterns)"""
        states = self.empty(count, 4)
        hashes = self.torch.empty(count, dtype=self.torch.int64, device=self.device)
        self.f.finish_candidates(step, first_slot, count, replies.data_ptr(), order.data_ptr(), targets.data_ptr(), states.data_ptr(),
                                 hashes.data_ptr())
        return states, hashes

    def kld_begin(self):
        self.f.kld_begin()

    def kld_feed(self, hashes):
        return self.f.kld_feed(hashes.data_ptr(), hashes.numel())

    def load_shard(self, states, shard_offset):
        self.f.load_shard(states.data_ptr(), states.shape[0], shard_offset)

    def sync(self):
        self.f.sync()

    def profile_enable(self, on=True):
        self.f.profile_enable(on)

    def profile_read(self, reset=True):
        return self.f.profile_read(reset)

    def close(self):
        self.f.close()


def shard_bounds(n_total: int, world: int, rank: int):
    """Contiguous, balanced split of the global index space [0, n_total)."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


class _Transport:
    """The collectives of the path.  RCCL takes device tensors directly; a gloo group (CPU ranks in the tests, or
    several ranks sharing one GPU) gets device tensors staged through host memory — same orchestration, slower wire."""

    def __init__(self, dist, group):
        self.dist, self.group = dist, group
        self.stage = dist.get_backend(group) == "gloo"

    def _host(self, t):
        return t.cpu() if (self.stage and t.is_cuda) else t

    def all_reduce_sum(self, t):
        h = self._host(t)
        self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
        if h is not t:
            t.copy_(h)

    def all_gather(self, out, t):
        ho, ht = self._host(out), self._host(t)
        self.dist.all_gather_into_tensor(ho, ht.contiguous(), group=self.group)
        if ho is not out:
            out.copy_(ho)

    def all_to_all(self, out, t, out_splits, in_splits):
        ho, ht = self._host(out), self._host(t)
        self.dist.all_to_all_single(ho, ht.contiguous(), out_splits, in_splits, group=self.group)
        if ho is not out:
            out.copy_(ho)


class ShardedAmcl:
    def __init__(self, grid, motion, sensor, params: AmclParams = AmclParams(), *, seed: int = 0, device: Optional[int] = None,
                 group=None, engine_factory=None, kld_block: Optional[int] = None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if not dist.is_initialized():
            raise RuntimeError("ShardedAmcl needs an initialised torch.distributed process group (one process per GPU)")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.params = params
        self.adaptive = params.min_particles < params.max_particles
        if self.adaptive and params.min_particles < self.world:
            raise ValueError("min_particles must be at least the number of ranks")
        self.net = _Transport(dist, group)
        self.kld_block = kld_block  # candidates drawn per round of C5 (doubles every round); None = max(min + 1, 8192 * world)
        self._pin_f64 = self._pin_i64 = self._pin_up = None  # pinned staging for the few scalars that cross to the host every cycle
        self.n_total = params.max_particles
        self.first_slot, self.n_local = shard_bounds(self.n_total, self.world, self.rank)
        if engine_factory is None:
            if device is None:
                device = torch.cuda.current_device()
            self.engine = HipShardEngine(grid, motion, sensor, params, seed, device, self.first_slot, self.n_local)
        else:
            self.engine = engine_factory(grid, motion, sensor, params, seed, self.first_slot, self.n_local)
        self.device = self.engine.device
        self._slow = _ExponentialFilter(params.alpha_slow)
        self._fast = _ExponentialFilter(params.alpha_fast)
        self._latest = None
        self._window = None
        self._every_n = 0
        self._force = True
        self._step = 0
        self._pivot = np.zeros(2)
        self.last_info = None
        self._initialized = False

    # -- reference surface -----------------------------------------------------------------------------
    def initialize(self, pose_xytheta, covariance):
        """Amcl::initialize(pose, covariance): every rank draws its slice of the same global sample stream."""
        self.n_total = self.params.max_particles
        self.first_slot, self.n_local = shard_bounds(self.n_total, self.world, self.rank)
        if self.adaptive:  # a KLD cut may have moved this shard: back to its slice of max_particles
            self.engine.load_shard(self.engine.empty(0, 4), self.first_slot)
        self.engine.initialize(pose_xytheta, covariance)
        self._pivot = np.array([float(pose_xytheta[0]), float(pose_xytheta[1])])
        self._force = True
        self._initialized = True

    def set_particles(self, states, weights):
        """Global (n_total x 4, n_total) arrays on every rank; each keeps its slice."""
        s = np.asarray(states, dtype=np.float64).reshape(-1, 4)
        w = np.asarray(weights, dtype=np.float64)
        assert len(w) == self.n_total
        sl = slice(self.first_slot, self.first_slot + self.n_local)
        self.engine.set_particles(s[sl], w[sl])
        self._pivot = np.array([s[0, 2], s[0, 3]])
        self._force = True
        self._initialized = True

    def particles(self):
        """This rank's shard (states, weights)."""
        return self.engine.particles()

    def gather_particles(self):
        """All shards, concatenated in global index order (debug / tests: moves N*40 bytes)."""
        s, w = self.engine.particles()
        objs = [None] * self.world
        self.dist.all_gather_object(objs, (s, w), group=self.group)
        return np.concatenate([o[0] for o in objs]), np.concatenate([o[1] for o in objs])

    def force_update(self):
        self._force = True

    def sync(self):
        self.engine.sync()

    def profile_enable(self, on=True):
        self.engine.profile_enable(on)

    def profile_read(self, reset=True):
        return self.engine.profile_read(reset)

    def update(self, control_action, measurement):
        """Amcl::update (amcl_core.hpp:165-201) over the sharded set. Returns (pose, covariance) or None."""
        scope = getattr(self.engine, "stream_scope", None)
        if scope is None:
            return self._update(control_action, measurement)
        with scope():
            return self._update(control_action, measurement)

    def _update(self, control_action, measurement):
        torch, dist = self.torch, self.dist
        if not self._initialized or self.n_total == 0:
            return None
        pose = np.asarray(control_action, dtype=np.float64)
        # on_motion (policies/on_motion.hpp:63-67,121-133)
        if self._latest is None:
            self._latest = pose.copy()
            moved = True
        else:
            d = _se2_mul(_se2_inverse(self._latest), pose)
            moved = math.hypot(d[2], d[3]) > self.params.update_min_d or abs(math.atan2(d[1], d[0])) > self.params.update_min_a
            if moved:
                self._latest = pose.copy()
        if not moved and not self._force:
            return None
        self._window = (pose.copy(), pose.copy()) if self._window is None else (pose.copy(), self._window[0])
        self._step += 1
        e = self.engine

        e.propagate(self._window[0], self._window[1], self._step)  # :174-175
        e.reweight(measurement)                                     # :176
        # every_n (every_n.hpp:47-50) does not depend on data: build the shard CDF only when it fires
        self._every_n = (self._every_n + 1) % self.params.resample_interval
        fires = self._every_n == 0
        # Scalars stay on the device between the kernels and the collectives; ONE host read-back for the policies.
        buf = e.empty(4)                     # [global weight sum | shard cdf total, shard sum w, shard sum w^2]
        buf.zero_()
        e.weight_sum_into(buf[0:1])
        self.net.all_reduce_sum(buf[0:1])                                           # C1
        e.normalize_from(buf[0:1], buf[2:4])                                        # :177 with the GLOBAL sum
        if fires:
            e.build_cdf_into(buf[1:2])
        gathered = e.empty(self.world * 3)
        self.net.all_gather(gathered, buf[1:4])                                     # C2
        both = e.empty(1 + self.world * 3)
        both[0:1] = buf[0:1]
        both[1:] = gathered
        host = self._to_host(both)
        weight_sum = float(host[0])
        stats = host[1:].reshape(self.world, 3)
        totals, norm_sum, norm_sumsq = stats[:, 0], float(stats[:, 1].sum()), float(stats[:, 2].sum())

        # :179 ThrunRecoveryProbabilityEstimator on the normalised weights (thrun_..._estimator.hpp:69-89)
        average = norm_sum / float(self.n_total)
        fast_average, slow_average = self._fast(average), self._slow(average)
        p_random = 0.0
        if abs(slow_average) >= np.finfo(np.float64).eps:
            p_random = min(max(1.0 - fast_average / slow_average, 0.0), 1.0)
        # :181 [&& on_effective_size_drop] (effective_sample_size.hpp:46-59)
        do_resample, ess = fires, -1.0
        if do_resample and self.params.selective_resampling:
            ess = 0.0 if norm_sum == 0.0 else norm_sum * norm_sum / norm_sumsq
            do_resample = ess < self.n_total * 0.5
        if do_resample:
            if p_random > 0.0:  # :184-186
                self._slow.reset()
                self._fast.reset()
            if self.adaptive:
                self._resample_kld(totals, p_random)
            else:
                self._resample(totals, p_random)
        self._force = False  # :199

        t_sums = e.empty(9)                                                 # :200
        e.estimate_sums_into(self._pivot, t_sums)
        self.net.all_reduce_sum(t_sums)                                     # C4
        sums = np.concatenate([self._to_host(t_sums), self._pivot, [0.0]])
        pose_est, cov = estimate_from_sums(sums)
        if np.all(np.isfinite(pose_est[2:])):
            self._pivot = np.array([pose_est[2], pose_est[3]])
        self.last_info = {"updated": True, "resampled": do_resample, "num_particles": self.n_total, "weight_sum": weight_sum,
                          "ess": ess, "random_state_probability": p_random}
        return pose_est, cov

    def _to_host(self, t):
        """Device tensor -> numpy through pinned memory (one asynchronous copy + one stream synchronisation)."""
        if not t.is_cuda:
            return t.numpy()
        torch = self.torch
        pin = self._pin_f64 if t.dtype == torch.float64 else self._pin_i64
        if pin is None or pin.numel() < t.numel():
            pin = torch.empty(max(256, t.numel()), dtype=t.dtype).pin_memory()
            if t.dtype == torch.float64:
                self._pin_f64 = pin
            else:
                self._pin_i64 = pin
        view = pin[:t.numel()]
        view.copy_(t.reshape(-1), non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return view.numpy().copy()

    def _to_device(self, host: np.ndarray):
        torch = self.torch
        if self.device.type != "cuda":
            return torch.from_numpy(np.ascontiguousarray(host, dtype=np.float64))
        # persistent pinned staging: the previous upload has completed (every cycle synchronises after it)
        if self._pin_up is None or self._pin_up.numel() < host.size:
            self._pin_up = torch.empty(max(64, host.size), dtype=torch.float64).pin_memory()
        stage = self._pin_up[:host.size]
        stage.copy_(torch.from_numpy(np.ascontiguousarray(host, dtype=np.float64)))
        return stage.to(self.device, non_blocking=True)

    def _cdf_intervals(self, totals: np.ndarray):
        ends_host = np.cumsum(totals)            # inclusive end of every shard's interval of the global CDF
        both = self._to_device(np.concatenate([ends_host, ends_host - totals]))
        return float(ends_host[-1]), both[:self.world], both[self.world:]

    def _draw(self, total, ends, offsets, p_random, first_slot, m):
        """Output slots [first_slot, first_slot + m) of views::sample | random_intersperse: the ancestor exchange (C3).
        -> (targets, replies (m, 4) records (x, y, cos, sin) in request order, order)"""
        torch, e, world = self.torch, self.engine, self.world
        targets = e.empty(m)
        e.resample_targets(self._step, p_random, total, first_slot, m, targets)
        # owner = first shard whose interval end is >= the target (std::lower_bound on the global CDF)
        requests_out, order, send_counts = e.route_targets(targets, ends, offsets, self.rank)
        all_counts = torch.empty(world * world, dtype=torch.int64, device=self.device)
        self.net.all_gather(all_counts, send_counts)                 # counts[r][q]: r asks q for that many
        counts = self._to_host(all_counts).reshape(world, world)
        send_list, recv_list = counts[self.rank].tolist(), counts[:, self.rank].tolist()
        requests_in = e.empty(int(sum(recv_list)))
        self.net.all_to_all(requests_in, requests_out, recv_list, send_list)
        served = e.serve_requests(requests_in)                      # (m_in, 4) records (x, y, cos, sin)
        replies = e.empty(m, 4)
        self.net.all_to_all(replies, served, send_list, recv_list)
        return targets, replies, order

    def _resample(self, totals: np.ndarray, p_random: float):
        """views::sample | random_intersperse | actions::assign (amcl_core.hpp:188-196) across shards (C3)."""
        total, ends, offsets = self._cdf_intervals(totals)
        m = self.n_local
        targets, replies, order = self._draw(total, ends, offsets, p_random, self.first_slot, m)
        self.engine.commit_routed(self._step, self.first_slot, m, replies, order, targets)

    def _resample_kld(self, totals: np.ndarray, p_random: float):
        """views::sample | random_intersperse | take_while_kld | take(max) | actions::assign (amcl_core.hpp:188-196, C5)."""
        torch, e, world, rank = self.torch, self.engine, self.world, self.rank
        total, ends, offsets = self._cdf_intervals(totals)
        max_p, min_p = self.params.max_particles, self.params.min_particles
        e.kld_begin()
        pos, block, blocks, n_out = 0, self.kld_block or max(min_p + 1, 8192 * world), [], max_p
        while pos < max_p:
            cnt = min(block, max_p - pos)
            lo, m = shard_bounds(cnt, world, rank)
            targets, replies, order = self._draw(total, ends, offsets, p_random, pos + lo, m)
            states, hashes = e.finish_candidates(self._step, pos + lo, m, replies, order, targets)
            # every rank checks kld_condition over the whole block, in global candidate order
            width, rem = -(-cnt // world), cnt % world
            if m < width:
                hashes = torch.cat([hashes, hashes.new_zeros(width - m)])
            gathered = torch.empty(world * width, dtype=torch.int64, device=self.device)
            self.net.all_gather(gathered, hashes)
            if rem:  # ranks >= rem hold one candidate less: drop their padding
                rows = gathered.view(world, width)
                gathered = torch.cat([rows[:rem].reshape(-1), rows[rem:, :width - 1].reshape(-1)])
            blocks.append((pos, cnt, states))
            fail = e.kld_feed(gathered)
            if fail is not None:
                n_out = fail        # the first candidate failing the predicate is dropped (take_while)
                break
            pos += cnt
            block *= 2
        n_out = min(n_out, max_p)   # | take(max)
        # re-balance: the kept candidates [0, n_out) become contiguous shards of the new set
        new_first, new_n = shard_bounds(n_out, world, rank)
        final = e.empty(new_n, 4)
        spans = [shard_bounds(n_out, world, r) for r in range(world)]

        def overlap(a0, a1, b0, b1):
            return max(0, min(a1, b1) - max(a0, b0))

        for pos, cnt, states in blocks:
            if pos >= n_out:
                break
            held = [(pos + lo, pos + lo + m) for lo, m in (shard_bounds(cnt, world, r) for r in range(world))]
            mine = held[rank]
            send = [overlap(mine[0], min(mine[1], n_out), f, f + c) for f, c in spans]
            recv = [overlap(h0, min(h1, n_out), new_first, new_first + new_n) for h0, h1 in held]
            # destinations are ordered by global index and cover [0, n_out): my kept candidates go out front to back,
            # and what I receive from this block is one contiguous run of my new shard, in source-rank = global order
            out0 = max(pos, new_first) - new_first
            self.net.all_to_all(final[out0:out0 + sum(recv)], states[:sum(send)], recv, send)
        e.load_shard(final, new_first)
        self.n_total, self.first_slot, self.n_local = n_out, new_first, new_n

    def close(self):
        self.engine.close()
