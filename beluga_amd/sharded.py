"""Particle-sharded MCL update across the GPUs of one node: one process per GPU, torch.distributed (RCCL) between them.

The reference is single-process (no collectives exist upstream).  The path shards naturally:
propagate / reweight are independent per particle; the couplings are
  C1  all-reduce of the weight sum (actions/normalize.hpp:70 becomes a global sum),
  C2  all-gather of {shard CDF total, sum w, sum w^2} (discrete_distribution's partial sums, ESS, Thrun average),
  C3  the ancestor exchange of multinomial resampling: every output slot j (global index space) draws
      u_j from the SAME Philox stream as the single-GPU path, finds the shard that owns that point of the global
      CDF, and fetches the ancestor's state from it (all-to-all of 8-byte targets out, 32-byte states back),
  C4  all-reduce of the nine estimate sums (algorithm/estimation.hpp:436-475).
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

    def weight_sum(self):
        return self.f.weight_sum()

    def normalize(self, factor):
        return self.f.normalize(factor)

    def build_cdf(self):
        return self.f.build_cdf()

    def resample_targets(self, step, p, total, first_slot, count, targets):
        self.f.resample_targets(step, p, total, first_slot, count, targets.data_ptr())

    def gather_by_cdf(self, targets, out4):
        m = targets.numel()
        self.f.gather_by_cdf(targets.data_ptr(), m, out4[0].data_ptr(), out4[1].data_ptr(), out4[2].data_ptr(), out4[3].data_ptr())

    def commit_resampled(self, step, first_slot, count, states4, targets):
        self.f.commit_resampled(step, first_slot, count, states4[0].data_ptr(), states4[1].data_ptr(), states4[2].data_ptr(),
                                states4[3].data_ptr(), targets.data_ptr())

    def estimate_sums(self, pivot):
        return self.f.estimate_sums(pivot)

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


class ShardedAmcl:
    def __init__(self, grid, motion, sensor, params: AmclParams = AmclParams(), *, seed: int = 0, device: Optional[int] = None,
                 group=None, engine_factory=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if not dist.is_initialized():
            raise RuntimeError("ShardedAmcl needs an initialised torch.distributed process group (one process per GPU)")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        if params.min_particles < params.max_particles:
            raise NotImplementedError("sharded KLD-adaptive resampling is not implemented yet; use min_particles == max_particles")
        self.params = params
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

    # -- collectives ---------------------------------------------------------------------------------
    def _all_reduce_sum(self, values):
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def _all_gather(self, values):
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.device)
        out = self.torch.empty(self.world * t.numel(), dtype=self.torch.float64, device=self.device)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().numpy().reshape(self.world, -1)

    # -- reference surface -----------------------------------------------------------------------------
    def initialize(self, pose_xytheta, covariance):
        """Amcl::initialize(pose, covariance): every rank draws its slice of the same global sample stream."""
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
        weight_sum = float(self._all_reduce_sum([e.weight_sum()])[0])  # C1
        local = e.normalize(weight_sum)                             # :177 with the GLOBAL sum; local sums of w, w^2 come back
        # every_n (every_n.hpp:47-50) does not depend on data: build the shard CDF only when it fires
        self._every_n = (self._every_n + 1) % self.params.resample_interval
        fires = self._every_n == 0
        cdf_total = e.build_cdf() if fires else 0.0
        stats = self._all_gather([cdf_total, local["norm_sum"], local["norm_sumsq"]])  # C2
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
            self._resample(totals, p_random)
        self._force = False  # :199

        sums = np.asarray(e.estimate_sums(self._pivot), dtype=np.float64)  # :200
        sums[:9] = self._all_reduce_sum(sums[:9].tolist())                 # C4
        pose_est, cov = estimate_from_sums(sums)
        if np.all(np.isfinite(pose_est[2:])):
            self._pivot = np.array([pose_est[2], pose_est[3]])
        self.last_info = {"updated": True, "resampled": do_resample, "num_particles": self.n_total, "weight_sum": weight_sum,
                          "ess": ess, "random_state_probability": p_random}
        return pose_est, cov

    def _resample(self, totals: np.ndarray, p_random: float):
        """views::sample | random_intersperse | actions::assign (amcl_core.hpp:188-196) across shards (C3)."""
        torch, dist, e = self.torch, self.dist, self.engine
        world, m = self.world, self.n_local
        ends_host = np.cumsum(totals)            # inclusive end of every shard's interval of the global CDF
        total = float(ends_host[-1])
        offsets = torch.tensor(ends_host - totals, dtype=torch.float64, device=self.device)
        ends = torch.tensor(ends_host, dtype=torch.float64, device=self.device)

        targets = e.empty(m)
        e.resample_targets(self._step, p_random, total, self.first_slot, m, targets)
        injected = torch.isnan(targets)
        lookup = torch.where(injected, torch.zeros_like(targets), targets)
        # owner = first shard whose interval end is >= the target (std::lower_bound on the global CDF)
        dest = torch.bucketize(lookup, ends, right=False).clamp_(max=world - 1)
        dest = torch.where(injected, torch.full_like(dest, self.rank), dest)
        local_targets = lookup - offsets[dest]
        order = torch.argsort(dest, stable=True)
        send_counts = torch.bincount(dest, minlength=world)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        send_list, recv_list = send_counts.tolist(), recv_counts.tolist()

        requests = e.empty(int(sum(recv_list)))
        dist.all_to_all_single(requests, local_targets[order].contiguous(), recv_list, send_list, group=self.group)
        served = e.empty(4, requests.numel())
        if requests.numel():
            e.gather_by_cdf(requests, served)
        replies = e.empty(m, 4)
        dist.all_to_all_single(replies, served.t().contiguous(), send_list, recv_list, group=self.group)
        states = e.empty(4, m)
        states[:, order] = replies.t()
        e.commit_resampled(self._step, self.first_slot, m, states, targets)

    def close(self):
        self.engine.close()
