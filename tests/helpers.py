"""Shared helpers for the parity tests: the same call sequence driven through the product
(stitching_amd, HIP) and through the oracle (oracle/oracle.py, CPU restatement)."""
import hashlib

import numpy as np

from stitching_amd import synthetic


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def small_ring(n, w, h, focal_factor=0.75, span=200.0):
    cams = synthetic.ring_cameras(n, w, h, focal_factor=focal_factor, span_deg=span)
    imgs = [synthetic.make_frame(i, w, h) for i in range(n)]
    return imgs, cams


def run_pipeline(mod_warper_cls, mod_blender_cls, imgs, cams, warper_type="spherical", blender_type="multiband",
                 blend_strength=5, masks_fn=None, aspect=1):
    """The linear warp -> blend sequence of the reference (stitching/verbose.py:77-95,176-181;
    tests/stitching_detailed.py:291-419) for either implementation."""
    warper = mod_warper_cls(warper_type)
    warper.set_scale(cams)
    sizes = [(im.shape[1], im.shape[0]) for im in imgs]
    w_imgs = [np.asarray(x) for x in warper.warp_images(imgs, cams, aspect)]
    w_masks = [np.asarray(x) for x in warper.create_and_warp_masks(sizes, cams, aspect)]
    corners, w_sizes = warper.warp_rois(sizes, cams, aspect)
    feed_masks = masks_fn(w_masks, corners, w_sizes) if masks_fn else w_masks
    blender = mod_blender_cls(blender_type, blend_strength)
    blender.prepare(corners, w_sizes)
    for img, mask, corner in zip(w_imgs, feed_masks, corners):
        blender.feed(img, mask, corner)
    pano, pmask = blender.blend()
    return dict(w_imgs=w_imgs, w_masks=w_masks, corners=corners, sizes=w_sizes, pano=np.asarray(pano),
                pmask=np.asarray(pmask), blender=blender)


class StripRecorder:
    """Transport stand-in, pass 1 of a sharded job executed rank after rank in ONE process: keeps what the rank sends (host
    copies) and hands back zero-filled buffers of the planned sizes (the bands of this pass are discarded)."""

    def __init__(self, ctx):
        self.ctx, self.sent, self.recvs = ctx, [], []

    def start(self, sends, recvs, ctx=None):
        self.sent = [(dst, np.asarray(p).reshape(-1)[:nb].copy()) for dst, p, nb in sends]
        self.recvs = recvs

    def finish(self, ctx=None):
        from stitching_amd.distributed import flat_device_buffer

        return [flat_device_buffer(self.ctx, np.zeros(nb, np.uint8)) for _, nb in self.recvs]


class StripReplay:
    """Pass 2: the strips the other ranks recorded arrive as this rank's incoming ones, in plan order."""

    def __init__(self, ctx, inbox):
        self.ctx, self.inbox, self.recvs = ctx, inbox, []

    def start(self, sends, recvs, ctx=None):
        self.recvs = recvs

    def finish(self, ctx=None):
        from stitching_amd.distributed import flat_device_buffer

        out = []
        for src, nb in self.recvs:
            a = self.inbox[src].pop(0)
            assert a.size == nb
            out.append(flat_device_buffer(self.ctx, a))
        return out


def run_sharded_job_in_one_process(ctx, frames, cams, world, per_rank, **job_kw):
    """Every rank of a ShardedStitchJob (as bench.py builds it for N > 1) executed one after the other on one GPU: pass 1
    records each rank's outgoing strips, pass 2 replays them as the incoming ones.  -> (panorama, mask, jobs): the
    concatenated bands of pass 2."""
    from stitching_amd.distributed import ShardedStitchJob

    jobs, recs = [], []
    for r in range(world):
        rec = StripRecorder(ctx)
        job = ShardedStitchJob(frames[r * per_rank:(r + 1) * per_rank], cams[r * per_rank:(r + 1) * per_rank], cams, r, world,
                               ctx=ctx, transport=rec, **job_kw)
        job.plan()
        out = job.run()
        del out
        jobs.append(job)
        recs.append(rec)
    bands = []
    for r in range(world):
        inbox = {src: [a for dst, a in recs[src].sent if dst == r] for src in range(world) if src != r}
        jobs[r].transport = StripReplay(ctx, inbox)
        bands.append(tuple(np.asarray(a) for a in jobs[r].run()))
    pano = np.concatenate([b[0] for b in bands], axis=1)
    mask = np.concatenate([b[1] for b in bands], axis=1)
    return pano, mask, jobs
