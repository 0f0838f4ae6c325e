"""Generates tests/golden/turtlebot3_world_grid.npz from the reference's example map (config 1 of BASELINE.json).

Run in the authoring container (the reference tree is not present on the GPU box):
    python tests/golden/make_turtlebot_grid.py
Reads  /root/reference/beluga_example/maps/turtlebot3_world.{pgm,yaml}
Writes the occupancy grid as nav_msgs/OccupancyGrid-style int8 (0 free / 100 occupied / -1 unknown) using
map_server's trinary rule (occupied_thresh 0.65, free_thresh 0.196, negate 0), rows flipped so that row 0 is
the bottom of the image (map_server convention), plus resolution and origin.
"""
import os
import re

import numpy as np

REF = "/root/reference/beluga_example/maps"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "turtlebot3_world_grid.npz")


def read_pgm(path):
    data = open(path, "rb").read()
    # P5 with optional comment lines
    tokens, pos = [], 0
    while len(tokens) < 4:
        m = re.compile(rb"\s*(#[^\n]*\n|\S+)").match(data, pos)
        pos = m.end()
        tok = m.group(1)
        if not tok.startswith(b"#"):
            tokens.append(tok)
    assert tokens[0] == b"P5"
    w, h, maxval = int(tokens[1]), int(tokens[2]), int(tokens[3])
    pos += 1  # single whitespace after maxval
    img = np.frombuffer(data, dtype=np.uint8, count=w * h, offset=pos).reshape(h, w)
    return img, maxval


def main():
    img, maxval = read_pgm(os.path.join(REF, "turtlebot3_world.pgm"))
    occ = (maxval - img.astype(np.float64)) / maxval  # negate: 0
    grid = np.full(img.shape, -1, dtype=np.int8)
    grid[occ > 0.65] = 100
    grid[occ < 0.196] = 0
    grid = np.ascontiguousarray(grid[::-1])  # image top row = max y
    np.savez_compressed(OUT, cells=grid, resolution=0.05, origin_xytheta=np.array([-10.0, -10.0, 0.0]))
    vals, counts = np.unique(grid, return_counts=True)
    print(OUT, dict(zip(vals.tolist(), counts.tolist())))


if __name__ == "__main__":
    main()
