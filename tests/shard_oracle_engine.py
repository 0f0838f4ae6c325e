"""A CPU stand-in for one shard's compute engine, built on the oracle (TEST INFRASTRUCTURE).

`beluga_amd.sharded.ShardedAmcl` drives a per-rank engine; on GPUs that engine is the HIP library.  Here the
same orchestration (collectives, CDF routing, ancestor exchange) runs on gloo/CPU ranks with every per-shard
kernel replaced by its oracle restatement, so the multi-rank logic can be tested without a GPU.
"""
import numpy as np
import torch

from oracle import binding as orc


class OracleShardEngine:
    def __init__(self, grid, motion, sensor, params, seed, shard_offset, shard_capacity):
        self.device = torch.device("cpu")
        self.grid, self.seed = grid, seed
        self.alphas = (motion.rotation_noise_from_rotation, motion.rotation_noise_from_translation,
                       motion.translation_noise_from_translation, motion.translation_noise_from_rotation)
        self.sensor = sensor
        self.lf = (sensor.max_obstacle_distance, sensor.max_laser_distance, sensor.z_hit, sensor.z_random, sensor.sigma_hit)
        self.field = orc.make_likelihood_field(grid.cells, grid.resolution, self.lf, sensor.model_unknown_space,
                                               sensor.only_obstacle_boundaries)
        self.offset, self.capacity = shard_offset, shard_capacity
        self.hash_res = (params.spatial_resolution_x, params.spatial_resolution_y, params.spatial_resolution_theta)
        self.kld = (params.min_particles, params.kld_epsilon, params.kld_z)
        self.states = np.zeros((0, 4))
        self.w = np.zeros(0)
        self.cdf = np.zeros(0)
        idx = np.flatnonzero(grid.cells.ravel() == grid.value_traits[0])
        W = grid.cells.shape[1]
        self.free_xy = np.stack([(idx % W + 0.5) * grid.resolution + grid.origin[2], (idx // W + 0.5) * grid.resolution + grid.origin[3]], 1)

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float64)

    def initialize(self, pose, cov):
        self.states, self.w = orc.init_normal(self.capacity, pose, cov, self.seed, index_offset=self.offset)

    def set_particles(self, states, weights):
        self.states, self.w = np.array(states, dtype=np.float64).reshape(-1, 4), np.array(weights, dtype=np.float64)

    def particles(self):
        return self.states.copy(), self.w.copy()

    def num_particles(self):
        return len(self.w)

    def propagate(self, pose, prev, step):
        sampler = orc.diffdrive_sampler(pose, prev, self.alphas)
        self.states = orc.propagate(self.states, sampler, self.seed, step, index_offset=self.offset)

    def reweight(self, points):
        self.w = self.w * orc.lf_weights(self.field, self.grid.resolution, self.grid.origin, self.lf[1], self.states, points)

    def weight_sum_into(self, t_sum):
        t_sum[0] = float(np.sum(self.w))

    def normalize_from(self, t_factor, t_stats2):
        f = float(t_factor[0])
        if not abs(f - 1.0) < np.finfo(np.float64).eps:
            self.w = self.w / f
        t_stats2[0] = float(np.sum(self.w))
        t_stats2[1] = float(np.sum(self.w * self.w))

    def build_cdf_into(self, t_total):
        self.cdf = np.cumsum(self.w)
        t_total[0] = float(self.cdf[-1]) if len(self.cdf) else 0.0

    def resample_targets(self, step, p, total, first_slot, count, targets):
        out = targets.numpy()
        for t in range(count):
            r = orc.draw(self.seed, step, 2, first_slot + t)
            inject = (first_slot + t) > 0 and p > 0.0 and (float(r[2]) * 2.0 ** -32) < p and len(self.free_xy) > 0
            u = float(((int(r[0]) << 32) | int(r[1])) >> 11) * 2.0 ** -53
            out[t] = np.nan if inject else u * total

    def route_targets(self, targets, ends, offsets, self_rank):
        t = targets.numpy()
        injected = np.isnan(t)
        lookup = np.where(injected, 0.0, t)
        dest = np.minimum(np.searchsorted(ends.numpy(), lookup, side="left"), len(ends) - 1)
        dest = np.where(injected, self_rank, dest)
        order = np.argsort(dest, kind="stable")
        local = np.where(injected, 0.0, lookup - offsets.numpy()[dest])
        counts = np.bincount(dest, minlength=len(ends)).astype(np.int64)
        return torch.from_numpy(local[order].copy()), torch.from_numpy(order.astype(np.int32)), torch.from_numpy(counts)

    def serve_requests(self, requests):
        idx = np.minimum(np.searchsorted(self.cdf, requests.numpy(), side="left"), len(self.cdf) - 1)
        s = self.states[idx]
        return torch.from_numpy(np.stack([s[:, 2], s[:, 3], s[:, 0], s[:, 1]], axis=1).copy())

    def commit_routed(self, step, first_slot, count, replies, order, targets):
        self.states, self.w = self._materialise(step, first_slot, count, replies, order, targets), np.ones(count)

    def _materialise(self, step, first_slot, count, replies, order, targets):
        r, t, o = replies.numpy(), targets.numpy(), order.numpy()
        new = np.zeros((count, 4))
        new[o] = np.stack([r[:, 2], r[:, 3], r[:, 0], r[:, 1]], axis=1)
        for k in np.flatnonzero(np.isnan(t)):
            d = orc.draw(self.seed, step, 3, first_slot + int(k))
            cell = min(int((float(((int(d[0]) << 32) | int(d[1])) >> 11) * 2.0 ** -53) * len(self.free_xy)), len(self.free_xy) - 1)
            theta = -np.pi + 2.0 * np.pi * (float(((int(d[2]) << 32) | int(d[3])) >> 11) * 2.0 ** -53)
            new[k] = orc.se2(self.free_xy[cell, 0], self.free_xy[cell, 1], theta)
        return new

    def finish_candidates(self, step, first_slot, count, replies, order, targets):
        new = self._materialise(step, first_slot, count, replies, order, targets)
        hashes = np.array([orc.spatial_hash(s, self.hash_res) for s in new], dtype=np.uint64)
        return torch.from_numpy(new), torch.from_numpy(hashes.view(np.int64).copy())

    def kld_begin(self):
        self.kld_hashes = np.zeros(0, dtype=np.uint64)

    def kld_feed(self, hashes):
        self.kld_hashes = np.concatenate([self.kld_hashes, hashes.numpy().view(np.uint64)])
        kept = orc.kld_take_while(self.kld_hashes, self.kld[0], self.kld[1], self.kld[2])
        return None if kept >= len(self.kld_hashes) else int(kept)

    def load_shard(self, states, shard_offset):
        self.states = states.numpy().copy().reshape(-1, 4)
        self.w = np.ones(len(self.states))
        self.offset = shard_offset

    def estimate_sums_into(self, pivot, t_sums9):
        w, s = self.w, self.states
        dx, dy = s[:, 2] - pivot[0], s[:, 3] - pivot[1]
        t_sums9.copy_(torch.tensor([w.sum(), (w * w).sum(), (w * s[:, 0]).sum(), (w * s[:, 1]).sum(), (w * dx).sum(), (w * dy).sum(),
                                    (w * dx * dx).sum(), (w * dx * dy).sum(), (w * dy * dy).sum()], dtype=torch.float64))

    def sync(self):
        pass

    def close(self):
        pass
