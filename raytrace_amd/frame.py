"""Frame-level driver: cut the image into tiles, give every GPU (rank) an interleaved share,
render each share with ONE kernel launch into one packed device buffer, and assemble the frame on
rank 0 with a single gather (RCCL over xGMI when the process group is `nccl`).

This replaces the reference's MPI scheme -- two row bands (80 % / 20 %) rendered redundantly by
half the ranks each with different seeds, sent with MPI_Isend / MPI_Recv to rank 0 and averaged
with 2/mpin (raytrace.cpp:179, 211, 219-227).  Here tiles are disjoint, so no averaging exists,
and because every path's random stream depends only on (seed, pixel, sub-pixel, sample) the
assembled frame is bit-identical for any tile size and any number of ranks.
"""
import numpy as np


def plan_tiles(sizex, sizey, tile_rows=64, tile_cols=64):
    """All tiles of a sizex (rows) x sizey (columns) image, row-major; edge tiles are ragged."""
    tiles = []
    for x0 in range(0, sizex, tile_rows):
        for y0 in range(0, sizey, tile_cols):
            tiles.append((x0, y0, min(tile_rows, sizex - x0), min(tile_cols, sizey - y0)))
    return tiles


def shard_tiles(tiles, rank, world):
    """Tile i belongs to rank i mod world (interleaved for load balance)."""
    return [t for i, t in enumerate(tiles) if i % world == rank]


def tile_pixels(tiles):
    return int(sum(t[2] * t[3] for t in tiles))


def dest_indices(tiles, sizey):
    """For a packed share (tiles back to back, row-major inside each tile): the linear pixel index
    x*sizey + y of every packed element, as int64."""
    out = []
    for (x0, y0, r, c) in tiles:
        out.append(((x0 + np.arange(r))[:, None] * sizey + (y0 + np.arange(c))[None, :]).reshape(-1))
    return np.concatenate(out).astype(np.int64) if out else np.zeros(0, np.int64)


def scatter_to_frame(frame, tiles, packed):
    """Copy a packed buffer (tiles back to back, row-major inside each tile) into frame[rows, cols, C].
    Works for numpy arrays and torch tensors alike."""
    off = 0
    for (x0, y0, r, c) in tiles:
        n = r * c
        frame[x0:x0 + r, y0:y0 + c] = packed[off:off + n].reshape(r, c, *frame.shape[2:])
        off += n
    return frame


class FrameRenderer:
    """One rank's view of a distributed frame render.

    `scene` is a DeviceScene on this rank's GPU.  `render()` launches this rank's tiles and, when
    world > 1, gathers all packed shares to rank 0, which scatters them into the full frame.
    """

    def __init__(self, scene, cam, params, tile_rows=64, tile_cols=64, rank=0, world=1, device=None, want_ids=False,
                 collective=None):
        """collective: run the gather through torch.distributed (default: when there is more than one rank; with one
        rank there is nothing to exchange, and a forced gather -- what the RCCL test does -- costs a copy of the frame)."""
        import torch
        self.torch = torch
        self.scene, self.cam, self.params = scene, cam, params
        self.rank, self.world = rank, world
        self.device = device if device is not None else torch.device("cuda", scene.dev)
        self.sizex, self.sizey = cam.sizex, cam.sizey
        self.all_tiles = plan_tiles(self.sizex, self.sizey, tile_rows, tile_cols)
        self.shards = [shard_tiles(self.all_tiles, r, world) for r in range(world)]
        self.my_tiles = self.shards[rank]
        self.counts = [tile_pixels(s) for s in self.shards]
        self.want_ids = want_ids
        if collective is None:
            collective = world > 1
        self.collective = bool(collective)
        # every rank's share is padded to the largest share so that one equal-size gather works
        self.max_pix = max(self.counts)
        self.local = torch.zeros((self.max_pix, 3), dtype=torch.float64, device=self.device)
        self.local_ids = torch.zeros((self.max_pix,), dtype=torch.int32, device=self.device) if want_ids else None
        self.gathered = None
        self.gathered_ids = None
        if self.collective and rank == 0:
            self.gathered = [torch.zeros_like(self.local) for _ in range(world)]
            if want_ids:
                self.gathered_ids = [torch.zeros_like(self.local_ids) for _ in range(world)]
        self.frame = torch.zeros((self.sizex, self.sizey, 3), dtype=torch.float64, device=self.device) if rank == 0 else None
        self.ids = torch.zeros((self.sizex, self.sizey), dtype=torch.int32, device=self.device) if (rank == 0 and want_ids) else None
        # un-permute tables (rank 0 only): packed position -> pixel index, one per rank
        self.dest = [torch.from_numpy(dest_indices(s, self.sizey)).to(self.device) for s in self.shards] if rank == 0 else None

    def launch(self):
        """Asynchronous: this rank's tiles, one kernel launch on torch's current stream."""
        if not self.my_tiles:
            return
        # the kernel goes on torch's current stream so that torch-side events and collectives order with it
        stream = self.torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None
        self.scene.render_tiles_device(self.cam, self.params, self.my_tiles, self.local.data_ptr(),
                                       self.local_ids.data_ptr() if self.want_ids else None, stream)

    def assemble(self):
        """Gather to rank 0 (one collective) and un-permute the tiles there. Returns the frame on rank 0."""
        torch = self.torch
        if self.collective:
            import torch.distributed as dist
            dist.gather(self.local, self.gathered if self.rank == 0 else None, dst=0)
            if self.want_ids:
                dist.gather(self.local_ids, self.gathered_ids if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        parts = self.gathered if self.collective else [self.local]
        flat = self.frame.view(-1, 3)
        for r in range(self.world):  # one index_copy per rank's share
            flat.index_copy_(0, self.dest[r], parts[r][: self.counts[r]])
        if self.want_ids:
            idparts = self.gathered_ids if self.collective else [self.local_ids]
            flat_ids = self.ids.view(-1)
            for r in range(self.world):
                flat_ids.index_copy_(0, self.dest[r], idparts[r][: self.counts[r]])
        return self.frame

    def render(self):
        self.launch()
        return self.assemble()


class FramePipeline:
    """Several frames of one scene in flight on this rank: frame i goes to slot i % depth, and every slot has its own
    FrameRenderer (share, gather and frame buffers) and its own stream, over ONE DeviceScene -- the library keeps two
    sets of per-launch state per scene, so launches on two streams overlap: one frame's last paths (a few long ones
    keep a few wavefronts busy) run beside the next frame's first, and a frame's gather beside the next frame's
    kernel.  A short frame -- one rank's share of a frame split eight ways -- gains most.  The frames are the same,
    bit for bit, as FrameRenderer's: a path's random stream depends only on (seed, pixel, sub-pixel, sample).

    With torch.distributed every rank must submit the same frames in the same order (each submit issues the slot's
    gather).  A slot's frame stays valid until the slot is submitted to again: take it with frame(slot) first."""

    def __init__(self, scene, cam, params, depth=2, **renderer_args):
        import torch
        self.torch = torch
        depth = max(1, int(depth))
        self.renderers = [FrameRenderer(scene, cam, type(params).from_buffer_copy(params), **renderer_args) for _ in range(depth)]
        self.device = self.renderers[0].device
        on_gpu = self.device.type == "cuda"
        if depth == 1 or not on_gpu:  # one frame at a time: the caller's stream
            self.streams = [None] * depth
        else:
            self.streams = [torch.cuda.Stream(self.device) for _ in range(depth)]
            for st in self.streams:  # whatever the caller queued before (uploads, clears) comes first
                st.wait_stream(torch.cuda.current_stream(self.device))
        self.depth = depth
        self.submitted = 0

    def _on(self, slot):
        import contextlib
        st = self.streams[slot]
        return self.torch.cuda.stream(st) if st is not None else contextlib.nullcontext()

    def submit(self, seed=None, kernel_events=None, assemble_events=None):
        """Queue the next frame (asynchronous) and return its slot.  kernel_events: a pair of torch events to record
        around the kernel, on the stream it runs on; assemble_events: likewise around the gather and the un-permute."""
        slot = self.submitted % self.depth
        self.submitted += 1
        fr = self.renderers[slot]
        if seed is not None:
            fr.params.seed = seed
        with self._on(slot):
            if kernel_events is not None:
                kernel_events[0].record()
            fr.launch()
            if kernel_events is not None:
                kernel_events[1].record()
            if assemble_events is not None:
                assemble_events[0].record()
            fr.assemble()
            if assemble_events is not None:
                assemble_events[1].record()
        return slot

    def frame(self, slot):
        """The slot's finished frame on rank 0 (None elsewhere); waits for it."""
        self.wait(slot)
        return self.renderers[slot].frame

    def wait(self, slot=None):
        for k in range(self.depth) if slot is None else [slot]:
            if self.streams[k] is not None:
                self.streams[k].synchronize()
            elif self.device.type == "cuda":
                self.torch.cuda.current_stream(self.device).synchronize()


def assemble_host(sizex, sizey, shards, packed_per_rank, channels=3, dtype=np.float64):
    """numpy version of the un-permute step, for checks without a GPU."""
    frame = np.zeros((sizex, sizey, channels) if channels else (sizex, sizey), dtype=dtype)
    for tiles, packed in zip(shards, packed_per_rank):
        scatter_to_frame(frame, tiles, np.asarray(packed)[: tile_pixels(tiles)])
    return frame
