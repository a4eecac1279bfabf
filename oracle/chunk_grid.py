"""Oracle twin of the host side of CUDASceneRepChunkGrid (TEST INFRASTRUCTURE
ONLY, like everything under oracle/; PARITY UNPINNED, see vh_oracle.h).

Restates, single-threaded and in float32, the host logic of
DepthSensingCUDA/Source/CUDASceneRepChunkGrid.{h,cpp}: streamOutToCPU (pass 0
on the oracle "GPU", pass 1 into the chunk grid), streamInToGPU (pass 0 picks
chunks entirely inside the sphere, pass 1 inserts them), the bit mask and the
chunk index arithmetic.  Pure Python: for the small scenes of the tests only.
"""
import math

import numpy as np

from voxelhashing_amd import vhtypes as T

f32 = np.float32


def _sign(v):
    return int(v > 0) - int(v < 0)


class OracleChunkGrid:
    def __init__(self, scene, voxel_extents, grid_dimensions, min_grid_pos, stream_out_parts):
        self.scene = scene
        self.ext = [f32(v) for v in voxel_extents]
        self.dims = [int(v) for v in grid_dimensions]
        self.min = [int(v) for v in min_grid_pos]
        self.max = [a + b for a, b in zip(self.min, self.dims)]
        self.parts = max(1, int(stream_out_parts))
        self.current_part = 0
        self.grid = {}  # chunk index -> list of (desc, block)
        n_bits = self.dims[0] * self.dims[1] * self.dims[2]
        self.bitmask = np.zeros((n_bits + 31) // 32, dtype=np.uint32)
        self.max_blocks = 100000  # CUDASceneRepChunkGrid.h:162

    # --- helpers, CUDASceneRepChunkGrid.h:560-614 -------------------------------------------
    def world_to_chunks(self, p):
        out = []
        for c, e in zip(p, self.ext):
            q = f32(f32(c) / e)
            out.append(int(np.trunc(f32(q + f32(_sign(q)) * f32(0.5)))))
        return out

    def chunk_to_world(self, c):
        return [f32(f32(ci) * e) for ci, e in zip(c, self.ext)]

    def is_valid_chunk(self, c):
        return all(self.min[i] <= c[i] < self.max[i] for i in range(3))

    def linearize(self, c):
        p = [(c[i] - self.min[i]) & 0xFFFFFFFF for i in range(3)]
        return (p[2] * self.dims[0] * self.dims[1] + p[1] * self.dims[0] + p[0]) & 0xFFFFFFFF

    def delinearize(self, idx):
        x = idx % self.dims[0]
        y = (idx % (self.dims[0] * self.dims[1])) // self.dims[0]
        z = idx // (self.dims[0] * self.dims[1])
        return [self.min[0] + x, self.min[1] + y, self.min[2] + z]

    def is_chunk_in_sphere(self, chunk, center, radius):
        # CUDASceneRepChunkGrid.h:317-346 (float32)
        pw = self.chunk_to_world(chunk)
        chunk_ext = max(self.ext)
        chunk_radius = f32(f32(f32(0.5) * chunk_ext) * f32(math.sqrt(f32(3.0))))
        d = [f32(pw[i] - f32(center[i])) for i in range(3)]
        l = f32(np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))))
        return bool(l <= abs(f32(f32(radius) - chunk_radius)))

    # --- stream out, CUDASceneRepChunkGrid.cpp:44-153 --------------------------------------------
    def stream_out_to_cpu(self, pos_camera, radius, use_parts=True):
        sc = self.scene
        sc.reset_mutex()
        ne = sc.num_entries()
        threads_per_part = (ne + self.parts - 1) // self.parts
        if not use_parts:
            threads_per_part = ne
        start = self.current_part * threads_per_part if use_parts else 0
        descs = sc.stream_out_pass1(threads_per_part, start, float(f32(radius)), [f32(v) for v in pos_camera], self.max_blocks)
        if use_parts:
            self.current_part = (self.current_part + 1) % self.parts
        if len(descs):
            blocks = sc.stream_out_pass2(descs)
            self.integrate_in_chunk_grid(descs, blocks)
        return len(descs)

    def integrate_in_chunk_grid(self, descs, blocks):
        vs = f32(self.scene.hp.m_virtualVoxelSize)
        for d, b in zip(descs, blocks):
            pw = [f32(f32(int(d["pos"][i]) * 8) * vs) for i in range(3)]
            chunk = self.world_to_chunks(pw)
            if not self.is_valid_chunk(chunk):
                continue
            idx = self.linearize(chunk)
            self.grid.setdefault(idx, []).append((d.copy(), b.copy()))
            self.bitmask[idx // 32] |= np.uint32(1 << (idx % 32))

    def stream_out_to_cpu_all(self):
        total = 1
        while total:
            total = 0
            for _ in range(self.parts):
                far = self.world_to_chunks([f32(m - 1) for m in self.min])
                total += self.stream_out_to_cpu([f32(v) for v in far], 0.0, True)

    # --- stream in, CUDASceneRepChunkGrid.cpp:197-311 ------------------------------------------------
    def stream_in_to_gpu(self, pos_camera, radius, use_parts=True):
        cam_chunk = self.world_to_chunks(pos_camera)
        rad = [int(math.ceil(f32(f32(radius) / e))) for e in self.ext]
        start = [max(cam_chunk[i] - rad[i], self.min[i]) for i in range(3)]
        end = [min(cam_chunk[i] + rad[i], self.max[i] - 1) for i in range(3)]
        descs, blocks = [], []
        done = False
        for x in range(start[0], end[0] + 1):
            for y in range(start[1], end[1] + 1):
                for z in range(start[2], end[2] + 1):
                    idx = self.linearize([x, y, z])
                    entries = self.grid.get(idx)
                    if not entries:
                        continue
                    if not self.is_chunk_in_sphere(self.delinearize(idx), pos_camera, radius):
                        continue
                    for d, b in entries:
                        descs.append(d)
                        blocks.append(b)
                    self.grid[idx] = []
                    self.bitmask[idx // 32] &= np.uint32(~np.uint32(1 << (idx % 32)))
                    if use_parts:
                        done = True
                        break
                if done:
                    break
            if done:
                break
        if descs:
            self.scene.stream_in(np.array(descs, dtype=T.DESC_DTYPE), np.stack(blocks))
        return len(descs)

    def stream_in_to_gpu_all(self, pos_camera, radius, use_parts=True):
        total, n = 0, 1
        while n:
            n = self.stream_in_to_gpu(pos_camera, radius, use_parts)
            total += n
        return total

    # --- content ---------------------------------------------------------------------------------------
    def host_blocks(self):
        """-> (descs, blocks) sorted by chunk index then block position"""
        descs, blocks = [], []
        for idx in sorted(self.grid):
            for d, b in self.grid[idx]:
                descs.append(d)
                blocks.append(b)
        if not descs:
            return np.zeros(0, dtype=T.DESC_DTYPE), np.zeros((0, T.SDF_BLOCK_VOXELS), dtype=T.VOXEL_DTYPE)
        return np.array(descs, dtype=T.DESC_DTYPE), np.stack(blocks)

    def statistics(self):
        return dict(chunks=sum(1 for v in self.grid.values() if v is not None),
                    blocks=sum(len(v) for v in self.grid.values()),
                    bits=int(np.unpackbits(self.bitmask.view(np.uint8)).sum()))
