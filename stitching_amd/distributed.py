"""Sharded warp -> blend over the GPUs of one node (one process per GPU).

The reference has no counterpart (single process, stitching/stitcher.py:247-254); this is the
MI355X-first design of SURVEY.md §8(e) / DESIGN.md §6:

  * images are dealt to ranks as contiguous runs (ring panoramas: contiguous yaw);
  * rank g owns the panorama columns [b_g, b_{g+1}) (edges on multiples of max(8, 2^bands));
  * every rank warps and builds pyramids for ITS images only;
  * an image whose 2^bands-aligned feed rectangle reaches another rank's columns sends that rank a strip.
    exchange="strips" (default): the COLUMNS of the warped image and mask the other band depends on (3 bytes per pixel
    + the mask as a byte or, 0 / 255 masks, as a bit; one pack kernel per 16 strips); the receiver feeds them like an
    image of its own and builds their pyramids.
    exchange="contribs" (round 1): per level the products (short)(L*W) and the weights W over the band's region
    (13.3 bytes per pixel and an export pass on the sender).  Either way this is the only data-path communication —
    point-to-point over xGMI (RCCL send/recv), never an all-reduce of the panorama pyramid;
  * the receiver adds everything in global feed order, so the assembled panorama is bit-identical to the single-GPU
    result.

The feather and the "no" blender (stitching/blender.py:27-36) shard the same way with simpler strips (round 3): both are per-pixel
once an image's weight map is known, and a feather weight min(dist * sharpness, 1) depends on the mask only within 1 / sharpness
columns — so the strip is the band's columns of the image plus that halo (none for "no"), the receiver feeds it like an image into a
blender prepared for its band (+ halo) and crops the result; everything is fed in global order (`flat_strip_columns`).

`ShardPlan` is pure geometry (every rank computes the same plan from the global camera list);
`ShardedStitchJob` runs one rank; transports move packed contribution buffers:
`RcclTransport` (C ABI -> librccl), `HostStagedTransport` (device -> host -> TCP -> device, for tests / 1-GPU boxes).

Control plane: a `group` object with the interface of `stitching_amd.rendezvous.TcpGroup` (plain sockets: the RCCL id broadcast,
one MIN vote, a plan-digest check, the gather of the bands).  No PyTorch anywhere in the package.
"""
import ctypes as C
import os

import numpy as np

from . import _lib, config
from .blender import Blender, _BlenderHandle
from .device import DeviceImage, as_device, get_context
from .stitching_error import StitchingError
from .synthetic import blend_strength_for_bands
from .warper import Warper


# ------------------------------------------------------------------------------------ blender handle
class ShardBlender(_BlenderHandle):
    """stx_blender with the sharding entry points of include/stitching_amd.h."""

    def set_band(self, x0, x1):
        _lib.check(self.ctx._lib.stx_blend_set_band(self._h, int(x0), int(x1)))

    def feed_ex(self, img, mask, corner, order):
        _lib.check(self.ctx._lib.stx_blend_feed_ex(self._h, img._h, mask._h, int(corner[0]), int(corner[1]), int(order)))

    def contrib_rect(self, size, corner, band):
        rect, nbytes = (C.c_int * 4)(), C.c_size_t()
        _lib.check(self.ctx._lib.stx_blend_contrib_rect(self._h, int(size[0]), int(size[1]), int(corner[0]),
                                                        int(corner[1]), int(band[0]), int(band[1]), rect,
                                                        C.byref(nbytes)))
        return tuple(int(v) for v in rect), int(nbytes.value)

    def export_contrib(self, order, band):
        out, rect = C.c_void_p(), (C.c_int * 4)()
        _lib.check(self.ctx._lib.stx_blend_export_contrib(self._h, int(order), int(band[0]), int(band[1]), C.byref(out),
                                                          rect))
        return DeviceImage(self.ctx, out), tuple(int(v) for v in rect)

    def export_contribs(self, items):
        """items: [(order, band)] -> [(packed DeviceImage, rect)]: all strips in one call, one launch per kernel
        instantiation instead of one per strip and level (stx_blend_export_contribs)."""
        n = len(items)
        if n == 0:
            return []
        orders = (C.c_int * n)(*[int(o) for o, _ in items])
        x0s = (C.c_int * n)(*[int(b[0]) for _, b in items])
        x1s = (C.c_int * n)(*[int(b[1]) for _, b in items])
        outs, rects = (C.c_void_p * n)(), (C.c_int * (4 * n))()
        _lib.check(self.ctx._lib.stx_blend_export_contribs(self._h, n, orders, x0s, x1s, outs, rects))
        return [(DeviceImage(self.ctx, C.c_void_p(outs[i])), tuple(int(v) for v in rects[4 * i:4 * i + 4])) for i in range(n)]

    def feed_strips(self, items, flags=_lib.CONTRIB_U8_BINARY):
        """items: [(flat buffer, w, h, corner, order)]: all received strips in one call (stx_blend_feed_strips)"""
        n = len(items)
        if n == 0:
            return
        bufs = (C.c_void_p * n)(*[i[0]._h for i in items])
        arr = lambda k: (C.c_int * n)(*[int(k(i)) for i in items])  # noqa: E731
        _lib.check(self.ctx._lib.stx_blend_feed_strips(self._h, n, bufs, arr(lambda i: i[1]), arr(lambda i: i[2]), arr(lambda i: i[3][0]),
                                                       arr(lambda i: i[3][1]), arr(lambda i: i[4]), int(flags)))

    def strip_rect(self, size, corner, band):
        """-> ((x0, x1) columns of the image, packed bytes) that an image owes the owner of `band`; x0 == x1: nothing."""
        xs, nbytes = (C.c_int * 2)(), C.c_size_t()
        _lib.check(self.ctx._lib.stx_strip_rect(self._h, int(size[0]), int(size[1]), int(corner[0]), int(corner[1]), int(band[0]),
                                                int(band[1]), xs, C.byref(nbytes)))
        return (int(xs[0]), int(xs[1])), int(nbytes.value)

    def build(self):
        """Build the pyramids of everything fed so far (otherwise deferred to the first export / blend())."""
        _lib.check(self.ctx._lib.stx_blend_build(self._h))

    def feed_contrib(self, order, rect, packed, flags=0):
        r = (C.c_int * 4)(*[int(v) for v in rect])
        _lib.check(self.ctx._lib.stx_blend_feed_contrib_ex(self._h, int(order), r, packed._h, int(flags)))


class GeometryContext:
    """Stands in for a device context when only the plan (band count, feed / contribution
    rectangles) is needed: blenders created on it cannot be fed."""

    handle = None
    device = -1

    def __init__(self):
        self._lib = _lib.lib()


def strip_pack(ctx, img, mask, x0, x1, flags=0):
    """columns [x0, x1) of a warped image + mask -> one flat device buffer (stx_strip_pack)"""
    return strip_pack_batch(ctx, [(img, mask, x0, x1)], flags)[0]


def strip_pack_batch(ctx, items, flags=0):
    """items: [(image, mask, x0, x1)] -> [flat device buffer]: every strip a rank owes, one copy kernel per 16 strips.
    flags: _lib.STRIP_MASK_BITS -> the masks (0 / 255 only) travel as one bit per pixel"""
    n = len(items)
    if n == 0:
        return []
    imgs = (C.c_void_p * n)(*[i[0]._h for i in items])
    masks = (C.c_void_p * n)(*[i[1]._h for i in items])
    x0s = (C.c_int * n)(*[int(i[2]) for i in items])
    x1s = (C.c_int * n)(*[int(i[3]) for i in items])
    outs = (C.c_void_p * n)()
    _lib.check(ctx._lib.stx_strip_pack_batch_ex(ctx.handle, n, imgs, masks, x0s, x1s, int(flags), outs))
    return [DeviceImage(ctx, C.c_void_p(outs[i])) for i in range(n)]


def strip_unpack(packed, w, h, flags=_lib.CONTRIB_U8_BINARY):
    """received flat buffer -> (image view, mask view)"""
    oi, om = C.c_void_p(), C.c_void_p()
    _lib.check(packed.ctx._lib.stx_strip_unpack(packed._h, int(w), int(h), int(flags), C.byref(oi), C.byref(om)))
    return DeviceImage(packed.ctx, oi), DeviceImage(packed.ctx, om)


def make_shard_blender(ctx, roi, num_bands):
    return ShardBlender(ctx or GeometryContext(), _lib.BLEND_MULTIBAND, num_bands, 0.0, roi)


# ------------------------------------------------------------------------------------ plan (geometry)
def owners_contiguous(n_images, world):
    """image k -> rank; contiguous runs, sizes differ by at most one."""
    base, extra = divmod(n_images, world)
    out = []
    for r in range(world):
        out += [r] * (base + (1 if r < extra else 0))
    return out


def band_edges(corners, sizes, owners, world, roi, num_bands):
    """Column bands [b_g, b_{g+1}) of the final roi, one per rank: the edge between two ranks sits
    midway between the centres of their images, snapped down to a multiple of max(8, 2^bands)."""
    align = max(8, 1 << num_bands)
    fw = roi[2]
    cx = [c[0] - roi[0] + s[0] / 2.0 for c, s in zip(corners, sizes)]
    means = []
    for g in range(world):
        mine = [cx[k] for k in range(len(cx)) if owners[k] == g]
        means.append(sum(mine) / len(mine) if mine else None)
    edges = [0]
    monotone = all(m is not None for m in means) and all(means[g] < means[g + 1] for g in range(world - 1))
    for g in range(1, world):
        if monotone:
            right_of_prev = max(cx[k] for k in range(len(cx)) if owners[k] == g - 1)
            left_of_next = min(cx[k] for k in range(len(cx)) if owners[k] == g)
            e = 0.5 * (right_of_prev + left_of_next)
        else:
            e = fw * g / world
        e = int(e) // align * align
        e = min(max(e, edges[-1] + align), fw - align * (world - g))
        edges.append(e)
    edges.append(fw)
    if any(edges[i + 1] <= edges[i] for i in range(world)):
        raise StitchingError(f"panorama of width {fw} is too narrow for {world} bands of >= {align} columns")
    return edges


FEATHER_DIST_CAP = 8192  # the distance transform saturates there (csrc/stx_blend.hip: 16-bit distances; OpenCV's 16.16 fixed point)


def feather_halo(sharpness):
    """Columns of mask a feather weight can depend on: weight = min(dist * sharpness, 1) with dist the L1 distance to the nearest
    zero of the image's mask (FeatherBlender::feed -> createWeightMap).  A zero farther than 1 / sharpness (or than the saturation
    distance) away leaves the weight at its cap, so with this many columns on both sides a strip's distance transform yields the
    weights of the whole image's wherever they are below the cap, and the cap elsewhere; + 2: fl(d * sharpness) >= 1 for every
    integer d beyond the halo, whatever the rounding of the product."""
    if not sharpness > 0:
        return FEATHER_DIST_CAP + 1
    return int(min(np.ceil(1.0 / float(np.float32(sharpness))) + 2, FEATHER_DIST_CAP + 1))


def flat_strip_columns(corner, size, roi, band, halo):
    """Columns [x0, x1) of an image that the owner of `band` needs under a per-pixel blender: where the image meets the band, widened
    by `halo` inside the image.  x0 == x1: the image does not reach the band."""
    tlx, w = int(corner[0]) - int(roi[0]), int(size[0])
    if min(band[1], tlx + w) <= max(band[0], tlx):
        return (0, 0)
    # the first column on a multiple of 8 (stx_strip_pack's rule: the mask bits of a row start on a byte); up to 7 more columns of halo
    return ((max(band[0] - halo, tlx) - tlx) & ~7, min(band[1] + halo, tlx + w) - tlx)


class ShardPlan:
    """Who owns which columns and which contribution strips travel where.  Pure geometry: built from
    the global corner/size lists, identical on every rank."""

    def __init__(self, corners, warped_sizes, owners, world, blender_probe, exchange="strips", mask_bits=False, kind="multiband",
                 halo=0, balance="midway"):
        """mask_bits (strips): the masks are known to hold 0 / 255 only and travel as one bit per pixel
        (STX_STRIP_MASK_BITS: 3.125 instead of 4 bytes per pixel on the links).
        kind: "multiband" (blender_probe: a ShardBlender of the panorama, asked for the strip geometry), or "feather" / "no"
        (blender_probe unused; halo: feather_halo(sharpness) / 0).
        balance: "midway" (band edges midway between the ranks' images) or "links" (edges moved towards equal widths as far as that
        lightens the busiest link)."""
        if exchange not in ("strips", "contribs"):
            raise StitchingError(f"unknown exchange form {exchange!r}")
        if kind not in ("multiband", "feather", "no"):
            raise StitchingError(f"unknown blender type {kind!r}")
        if kind != "multiband" and exchange != "strips":
            raise StitchingError("the feather and the plain blender exchange image strips")
        self.kind, self.halo = kind, int(halo) if kind == "feather" else 0
        self.exchange = exchange
        self.mask_bits = bool(mask_bits) and exchange == "strips"
        self.strip_flags = _lib.STRIP_MASK_BITS if self.mask_bits else 0
        self.corners = [tuple(int(v) for v in c) for c in corners]
        self.sizes = [tuple(int(v) for v in s) for s in warped_sizes]
        self.owners = list(owners)
        self.world = int(world)
        self.roi = Blender.result_roi(self.corners, self.sizes)
        self.num_bands = blender_probe.num_bands() if kind == "multiband" else 0
        self._probe = blender_probe
        self.balance = balance
        self.edges = band_edges(self.corners, self.sizes, self.owners, self.world, self.roi, self.num_bands)
        self.messages = self._messages()
        if balance == "links" and self.world > 2:
            # Band edges between "midway between the ranks' images" (even pyramid work, but the two end bands of a multi-row panorama
            # take in the overhang of their wide frames and the end links carry it) and equal widths (the reverse): the mix with the
            # lightest busiest link.  Config 3: 96.5 MB on the links 1 -> 0 and 6 -> 7 with the midway edges, 77.8 MB a quarter of the
            # way to equal widths — for 9 % more pyramid samples on the busiest rank.  Pure geometry: every rank
            # evaluates the same candidates in the same order.
            align, fw = max(8, 1 << self.num_bands), self.roi[2]
            midway = list(self.edges)
            best = (self.busiest_link_bytes(), midway, self.messages)
            for t in (0.125, 0.25, 0.375, 0.5):
                cand = [0]
                for g in range(1, self.world):
                    e = int((1.0 - t) * midway[g] + t * fw * g / self.world) // align * align
                    cand.append(min(max(e, cand[-1] + align), fw - align * (self.world - g)))
                cand.append(fw)
                self.edges = cand
                msgs = self._messages()
                self.messages = msgs
                load = self.busiest_link_bytes()
                if load < best[0]:
                    best = (load, cand, msgs)
            _, self.edges, self.messages = best
        elif balance not in ("midway", "links"):
            raise StitchingError(f"unknown band balance {balance!r}")

    def _messages(self):
        """(order k, src rank, dst rank, rect, bytes) of every strip under the current edges, sorted by (dst, k) so that every rank posts
        sends / receives in one global order.  rect: contribs -> (x, y, w, h) relative to the roi; strips -> (x0, x1, w, h): columns
        of image k and the strip's size"""
        out = []
        for k, (c, s) in enumerate(zip(self.corners, self.sizes)):
            for g in range(self.world):
                if g == self.owners[k]:
                    continue
                if self.exchange == "strips":
                    if self.kind == "multiband":
                        (x0, x1), nbytes = self._probe.strip_rect(s, c, self.band(g))
                    else:
                        (x0, x1), nbytes = flat_strip_columns(c, s, self.roi, self.band(g), self.halo), None
                    if x1 > x0:
                        if self.mask_bits or nbytes is None:
                            nb = C.c_size_t()
                            _lib.check(_lib.lib().stx_strip_bytes(x1 - x0, int(s[1]), self.strip_flags, C.byref(nb)))
                            nbytes = int(nb.value)
                        out.append((k, self.owners[k], g, (x0, x1, x1 - x0, s[1]), nbytes))
                    continue
                rect, nbytes = self._probe.contrib_rect(s, c, self.band(g))
                if rect[2] > 0:
                    out.append((k, self.owners[k], g, rect, nbytes))
        out.sort(key=lambda m: (m[2], m[0]))
        return out

    def link_bytes(self):
        """{(src, dst): bytes per panorama}: xGMI is point to point, so these are per-link loads"""
        links = {}
        for (_k, src, dst, _rect, nbytes) in self.messages:
            links[(src, dst)] = links.get((src, dst), 0) + nbytes
        return links

    def busiest_link_bytes(self):
        return max(self.link_bytes().values(), default=0)

    def band(self, g):
        return (self.edges[g], self.edges[g + 1])

    def band_roi(self, g):
        """feather / no: (roi the blender of rank g is prepared for — its band + halo, absolute panorama coordinates —, the band's
        columns inside it)"""
        b0, b1 = self.band(g)
        lo, hi = max(b0 - self.halo - 7, 0), min(b1 + self.halo, self.roi[2])  # - 7: flat_strip_columns rounds a strip's start down
        return (self.roi[0] + lo, self.roi[1], hi - lo, self.roi[3]), (b0 - lo, b1 - lo)

    def own_columns(self, k, g):
        """feather / no: the columns of image k that rank g's blender is fed (its own image or a received strip)"""
        return flat_strip_columns(self.corners[k], self.sizes[k], self.roi, self.band(g), self.halo)

    def sends(self, rank):
        return [m for m in self.messages if m[1] == rank]

    def recvs(self, rank):
        return [m for m in self.messages if m[2] == rank]

    def exchanged_bytes(self):
        return sum(m[4] for m in self.messages)


# ------------------------------------------------------------------------------------ transports
class HostStagedTransport:
    """Host-staged exchange over the control-plane group (rendezvous.TcpGroup or anything with its `exchange_bytes`):
    device -> host -> peer -> device.  Works with several ranks on ONE GPU and on CPU-only rendezvous; used by the tests and as
    the fallback when RCCL cannot initialise."""

    name = "host-staged"

    def __init__(self, group, ctx):
        self.group, self.ctx = group, ctx
        self._pending = {}  # id(context) -> (sends, recvs, context): one exchange per context may be pending, as with RcclTransport
        self._last = None

    def exchange(self, sends, recvs, ctx=None):
        """sends: [(dst, DeviceImage packed, nbytes)], recvs: [(src, nbytes)] -> [DeviceImage]"""
        host_sends = [(dst, np.asarray(packed).reshape(-1)[:nbytes]) for dst, packed, nbytes in sends]
        return [flat_device_buffer(ctx or self.ctx, a) for a in self.group.exchange_bytes(host_sends, recvs)]

    # split form (same contract as RcclTransport): the host-staged exchange has nothing to overlap, it runs in finish()
    # Several contexts (panoramas in flight) may share the transport; the exchanges then run in the order of the finish()
    # calls, which must be the same on every rank.
    def start(self, sends, recvs, ctx=None):
        ctx = ctx or self.ctx
        if id(ctx) in self._pending:
            raise StitchingError("an exchange of this context is already pending on this transport")
        self._pending[id(ctx)] = (sends, recvs, ctx)
        self._last = ctx

    def finish(self, ctx=None):
        ctx = ctx or self._last
        if ctx is None or id(ctx) not in self._pending:
            raise StitchingError("finish() without a pending exchange of this context")
        sends, recvs, ctx = self._pending.pop(id(ctx))
        return self.exchange(sends, recvs, ctx)


def flat_device_buffer(ctx, host_bytes):
    """1-D uint8 host array -> flat device buffer with the geometry stx_blend_export_contrib uses."""
    n = int(host_bytes.size)
    w = min(n, 1 << 30)
    h = (n + (1 << 30) - 1) >> 30
    if h != 1:
        raise StitchingError("contribution larger than 1 GiB is not supported by the host-staged transport")
    return DeviceImage.from_numpy(host_bytes.reshape(1, w), ctx)


class RcclTransport:
    """RCCL send/recv of the packed contribution buffers on the context's HIP stream
    (stx_comm_* in include/stitching_amd.h).  The unique id is distributed by the control plane
    (default_transport: one broadcast over the rendezvous group)."""

    name = "rccl"

    def __init__(self, ctx, rank, world, unique_id):
        self.ctx, self.rank, self.world = ctx, rank, world
        h = C.c_void_p()
        uid = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        _lib.check(ctx._lib.stx_comm_create(ctx.handle, int(world), int(rank), uid, C.byref(h)))
        self._h = h
        self._inflight, self._last = {}, None

    def info(self):
        """What librccl itself says about this communicator: ranks it counts, this rank, its version (e.g. 22203), the device"""
        out = (C.c_int * 4)()
        _lib.check(self.ctx._lib.stx_comm_info(self._h, out))
        return {"rccl_ranks": int(out[0]), "rccl_user_rank": int(out[1]), "rccl_version": int(out[2]), "device": int(out[3])}

    @staticmethod
    def unique_id():
        """ncclGetUniqueId.  No environment is touched here: a caller whose ranks all share one host may wrap this call and the
        constructor in `loopback_bootstrap()` (default_transport does, after comparing host names)."""
        uid = (C.c_ubyte * 128)()
        _lib.check(_lib.lib().stx_comm_unique_id(uid))
        return bytes(uid)

    def start(self, sends, recvs, ctx=None):
        """Issue the exchange on the communicator's stream (after everything queued so far on the stream of `ctx`,
        default: the communicator's context).  Kernels launched on that stream until finish() overlap with the
        transfer.  Several contexts (panoramas in flight) may share one transport: call start/finish pairs in the
        same order on every rank."""
        ctx = ctx or self.ctx
        lib = ctx._lib
        rbufs = []
        n = len(sends) + len(recvs)
        peers, is_send = (C.c_int * max(n, 1))(), (C.c_int * max(n, 1))()
        ptrs, sizes = (C.c_void_p * max(n, 1))(), (C.c_size_t * max(n, 1))()
        i = 0
        for src, nbytes in recvs:
            out = C.c_void_p()
            _lib.check(lib.stx_buf_alloc(ctx.handle, min(nbytes, 1 << 30), (nbytes + (1 << 30) - 1) >> 30, 1, _lib.U8,
                                         C.byref(out)))
            buf = DeviceImage(ctx, out)
            rbufs.append(buf)
            peers[i], is_send[i], ptrs[i], sizes[i] = src, 0, buf.device_ptr(), nbytes
            i += 1
        for dst, packed, nbytes in sends:
            if packed.ctx is not ctx:
                raise StitchingError("strips must be packed on the context that exchanges them (its stream orders their lifetime)")
            peers[i], is_send[i], ptrs[i], sizes[i] = dst, 1, packed.device_ptr(), nbytes
            i += 1
        _lib.check(lib.stx_comm_exchange_begin_on(self._h, ctx.handle, n, peers, is_send, ptrs, sizes))
        self._inflight[id(ctx)] = (rbufs, [p for _, p, _ in sends], ctx)  # both sides stay alive until finish()
        self._last = ctx

    def finish(self, ctx=None):
        """Order the stream of `ctx` (default: the context of the latest start) after ITS transfer — exchanges of
        other contexts that were started in between are not waited for; returns the received strips."""
        rbufs, sent, ctx = self._inflight.pop(id(ctx or self._last))
        # stx_comm_exchange_end_on makes the context's stream wait for the event recorded on the communicator's stream
        # behind the whole send / recv group.  The sent strips were allocated by that context, whose allocator is ordered by
        # its stream: once the wait is queued, a block released here can only be handed to work that runs after the
        # transfer — so the references are dropped now (no "keep n generations" guess).
        _lib.check(ctx._lib.stx_comm_exchange_end_on(self._h, ctx.handle))
        del sent
        return rbufs

    def exchange(self, sends, recvs, ctx=None):
        self.start(sends, recvs, ctx)
        return self.finish()

    def close(self):
        if self._h is not None:
            if self.ctx.handle:  # stx_comm_destroy uses the context's device and stream: never after Context.close()
                self.ctx._lib.stx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------ one rank
class ShardedStitchJob:
    """One rank of a sharded panorama: device-resident local frames + the global camera list."""

    def __init__(self, frames, cameras, all_cameras, rank, world, all_sizes=None, warper_type="spherical",
                 blender_type="multiband", num_bands=5, blend_strength=None, ctx=None, group=None, transport=None,
                 split_boundary=True, exchange="strips", mask_bits=True, balance=None, dist=None, compensator=None, seam_masks=None):
        """blender_type / blend_strength: as stitching.blender.Blender (stitching/blender.py:5-38); for "multiband" `num_bands` sets the
        band count when blend_strength is None (the benchmark's way of naming a configuration).
        split_boundary (multiband): warp / feed the images that owe strips to other ranks first and the rest while the strips
        travel (lowest latency of ONE panorama).  A caller that keeps several panoramas in flight on several contexts
        passes False: all local images go through one warp launch and one pyramid build, and the other panorama's
        kernels fill the time of the exchange.
        exchange: "strips" (warped-image columns) or, multiband only, "contribs" (round 1's per-level products).
        balance: ShardPlan's band-edge rule, "links" (default) or "midway".
        group: the control plane (stitching_amd.rendezvous.TcpGroup or an object with its interface; `dist` is the old name of the
        argument): transport negotiation, the plan check of plan(), gather().  A job that is handed a transport and never gathers
        needs none.
        compensator: an ExposureErrorCompensator whose gains are indexed by the GLOBAL image order (every rank holds the same one, as
        every rank holds all cameras): applied to this rank's warped images between warp and feed (stitching/stitcher.py:123) — the strips
        a rank sends are strips of compensated images, so the assembled panorama equals the single-GPU one.
        seam_masks: LOW-resolution seam masks by global image order (every rank holds all of them: a few KB each), resized on the device to
        each warped mask and ANDed with it (SeamFinder.resize, stitching/stitcher.py:124) before the image is fed or its strips are cut —
        the masks are then grey along the seams and travel as bytes."""
        if blender_type not in Blender.BLENDER_CHOICES:
            raise StitchingError(f"unknown blender type {blender_type!r}")
        if blender_type != "multiband" and exchange != "strips":
            raise StitchingError("the feather and the plain blender exchange image strips")
        self.blender_type = blender_type
        self.ctx = ctx or get_context()
        self.rank, self.world = int(rank), int(world)
        self.frames = [as_device(f, self.ctx) for f in frames]
        self.cameras = list(cameras)
        self.all_cameras = list(all_cameras)
        n = len(self.all_cameras)
        self.owners = owners_contiguous(n, self.world)
        self.my_orders = [k for k in range(n) if self.owners[k] == self.rank]
        if len(self.my_orders) != len(self.frames):
            raise StitchingError(f"rank {rank} holds {len(self.frames)} frames but owns {len(self.my_orders)} images")
        size0 = (self.frames[0].width, self.frames[0].height)
        self.all_sizes = list(all_sizes) if all_sizes is not None else [size0] * n
        self.warper = Warper(warper_type, ctx=self.ctx)
        self.warper.set_scale(self.all_cameras)
        self._cam_arrays = self.warper.camera_arrays(self.cameras)  # K, R of this rank's cameras as the batched entry points take them
        self.num_bands_req, self.blend_strength = num_bands, blend_strength
        self.compensator = compensator
        self._local_comp = {}
        self.seam_masks = None if seam_masks is None else list(seam_masks)
        self._seam_dev = {}
        self._seam_version = 0
        self._agreed_inputs = None  # (gains version, seam-mask version) the ranks last compared digests under
        self.dist = group if group is not None else dist  # `dist`: the older name of the same argument
        if self.dist is not None:
            missing = [m for m in ("all_gather", "gather", "broadcast", "barrier", "all_reduce_min", "exchange_bytes") if not hasattr(self.dist, m)]
            if missing:
                raise StitchingError(f"group= needs the TcpGroup interface (stitching_amd/rendezvous.py); {type(self.dist).__name__} lacks "
                                     f"{', '.join(missing)} — wrap it like tests/gloo_group.py wraps torch.distributed")
        self.transport = transport
        self.split_boundary = bool(split_boundary)
        self.exchange = exchange
        # the masks of a job without seam masks are warped masks (0 / 255): they may travel as bits; resized seam masks are grey
        self.mask_bits = bool(mask_bits) and seam_masks is None
        self._bin_flag = _lib.CONTRIB_U8_BINARY if seam_masks is None else 0
        # ShardPlan(balance=...): where the band edges go; None -> STITCHING_AMD_BALANCE, else "links".  Measured on one MI355X playing
        # single ranks of the 8-rank config-3 job (tools/sim_rank.py, gpurun r3o): rank 3 0.834 -> 0.874 ms per step, rank 0 0.776 ->
        # 0.680, rank 1 0.723 -> 0.730, while the job's busiest link goes from 96.5 to 77.8 MB per panorama
        self.balance = balance or os.environ.get("STITCHING_AMD_BALANCE") or "links"
        self.plan_ = None

    @property
    def source_pixels(self):
        return sum(f.width * f.height for f in self.frames)

    def plan(self):
        refusal = None  # a reason this rank cannot run the job, raised on every rank together after the agreement round
        corners, wsizes = self.warper.warp_rois(self.all_sizes, self.all_cameras)
        roi = Blender.result_roi(corners, wsizes)
        self.roi = roi
        flat = self.blender_type != "multiband"
        if self.blend_strength is None:
            self.blend_strength = Blender.DEFAULT_BLEND_STRENGTH if flat else blend_strength_for_bands(self.num_bands_req, roi[2], roi[3])
        blend_width = np.sqrt(roi[2] * roi[3]) * self.blend_strength / 100
        if flat or blend_width < 1:
            # Blender.prepare's choice (stitching/blender.py:25-36): "no", or a blend width below one pixel -> the plain blender,
            # whatever type was asked for
            if self.blender_type == "feather" and blend_width >= 1:
                self.flat_kind, self.sharpness = "feather", 1.0 / blend_width
            else:
                self.flat_kind, self.sharpness = "no", 0.0
            self.req_bands = 0
            self.plan_ = ShardPlan(corners, wsizes, self.owners, self.world, None, "strips", self.mask_bits, kind=self.flat_kind,
                                   halo=feather_halo(self.sharpness), balance=self.balance)
        else:
            # pyr_order() (csrc/stx_blend.hip) splits a row into vector body and scalar tail from the width and x of the FEED
            # RECTANGLE; an exchange strip's rectangle is not its image's, so under a vector-order model the fp32 weights of a
            # strip and of the whole image may round differently at the same column: "sharded == single GPU, bit for bit" would
            # not hold.  The models exist to be compared with on one GPU; sharded jobs refuse them — AFTER the
            # ranks have compared notes (_check_plan_agreement), so that a rank whose environment alone names such a mode does not
            # leave its peers waiting in the agreement's all_gather until the rendezvous times out.
            if config.pyrdown_mode()[0] != "scalar":
                refusal = "sharded multi-band blending needs the scalar pyrDown order (STITCHING_AMD_PYRDOWN / set_pyrdown_mode)"
            self.req_bands = int((np.log(blend_width) / np.log(2.0) - 1.0))
            probe = make_shard_blender(self.ctx, roi, self.req_bands)
            self.plan_ = ShardPlan(corners, wsizes, self.owners, self.world, probe, self.exchange, self.mask_bits, balance=self.balance)
        self.last_num_bands = self.plan_.num_bands
        self._check_plan_agreement(refusal)
        self._local_comp, self._seam_dev = {}, {}
        self._agreed_inputs = self._inputs_version()
        if self.transport is None:
            self.transport = default_transport(self.ctx, self.rank, self.world, self.dist)
        return self.plan_

    def set_seam_masks(self, seam_masks):
        """New low-resolution seam masks (by global image order) for the following run()s: call it on EVERY rank between the same two
        runs — the next run() compares digests again before a strip moves."""
        self.seam_masks = None if seam_masks is None else list(seam_masks)
        self._seam_dev = {}
        self._seam_version += 1

    def _inputs_version(self):
        gv = getattr(self.compensator, "gains_version", 0) if self.compensator is not None else 0
        # the gains and seam masks are also compared by identity: a caller that assigned `.gains` / `.seam_masks` directly is still noticed
        ids = (id(getattr(self.compensator, "gains", None)), tuple(id(m) for m in self.seam_masks) if self.seam_masks is not None else None)
        return (gv, self._seam_version, ids)

    def _refresh_inputs(self):
        """Gains (compensator.set_gains, e.g. re-estimated while a stream runs) or seam masks (set_seam_masks) changed since the ranks
        last agreed: drop everything derived from the old ones — this rank's per-image compensators, the uploaded seam masks — and
        compare digests again.  The agreement is a collective: set_gains / set_seam_masks must be called on every rank between the same
        two run()s (SPMD); a rank that was left out keeps its peers in the all_gather until the group's timeout names it."""
        v = self._inputs_version()
        if v == self._agreed_inputs:
            return
        self._local_comp = {}
        self._seam_dev = {}
        if self._agreed_inputs is not None:
            self._check_plan_agreement()
        self._agreed_inputs = v

    def plan_digest(self):
        """What every rank must agree on before the first strip moves: the geometry (edges, messages) and everything process-wide that
        changes a rank's ROIs or bytes (band balance, trig / remap / pyrDown modes)."""
        import hashlib

        p = self.plan_
        comp = None
        if self.compensator is not None and self.compensator.compensator_type != "no" and self.compensator.gains is not None:
            # a rank with other gains would not hang anybody, it would quietly blend other bytes: the gains belong to the agreement
            h = hashlib.sha256()
            for g in self.compensator.gains:
                h.update(np.ascontiguousarray(g).tobytes())
            comp = (self.compensator.compensator_type, h.hexdigest())
        if self.seam_masks is not None:
            h = hashlib.sha256()
            for m in self.seam_masks:
                a = np.ascontiguousarray(np.asarray(m.get() if hasattr(m, "get") else m))
                h.update(repr(a.shape).encode() + a.tobytes())
            comp = (comp, "seams", h.hexdigest())
        state = (p.kind, p.exchange, p.mask_bits, p.halo, p.num_bands, p.balance, list(p.edges), [tuple(m) for m in p.messages],
                 list(p.corners), list(p.sizes), config.trig_mode(), config.remap_mode(), tuple(config.pyrdown_mode()),
                 float(self.blend_strength), float(getattr(self, "sharpness", 0.0)), comp)
        return hashlib.sha256(repr(state).encode()).hexdigest()

    def _check_plan_agreement(self, refusal=None):
        """ShardPlan is "pure geometry, identical on every rank" only while every rank sees the same cameras AND the same environment
        (STITCHING_AMD_BALANCE, the arithmetic modes).  A rank that differs would post other sends / receives than its peers expect and
        the job would hang in the exchange: compare digests over the control plane first and fail with a message instead."""
        if self.world == 1 or self.dist is None:
            if refusal:
                raise StitchingError(refusal)
            return
        mine = self.plan_digest()
        both = self.dist.all_gather((mine, refusal))
        all_, refused = [b[0] for b in both], [(r, b[1]) for r, b in enumerate(both) if b[1]]
        if refused:  # every rank raises, together
            raise StitchingError(f"rank {self.rank}: " + "; ".join(f"rank {r}: {msg}" for r, msg in refused))
        if len(set(all_)) != 1:
            odd = [r for r, d in enumerate(all_) if d != all_[0]]
            raise StitchingError(f"rank {self.rank}: the shard plan differs between ranks (ranks {odd} disagree with rank 0): cameras, "
                                 "STITCHING_AMD_BALANCE or the trig / remap / pyrDown modes are not the same in every process")

    def run(self):
        """warp + feed local images, exchange contribution strips, blend this rank's band.
        Returns device-resident (band u8x3, band mask u8)."""
        p = self.plan_ or self.plan()
        self._refresh_inputs()
        # this rank's share of the ROI pass belongs to every panorama (as in StitchJob.run)
        local = {k: i for i, k in enumerate(self.my_orders)}
        corners, _ = self.warper.warp_rois([self.all_sizes[k] for k in self.my_orders], self.cameras, camera_arrays=self._cam_arrays)
        if [tuple(c) for c in corners] != [p.corners[k] for k in self.my_orders]:
            raise StitchingError("warp rois changed between plan() and run()")
        prev = config.device_resident()
        config.set_device_resident(True)
        if p.kind != "multiband":
            try:
                return self._run_flat(p)
            finally:
                config.set_device_resident(prev)
        try:
            if config.pyrdown_mode()[0] != "scalar":  # the mode is read again when a blender builds its pyramids: it must still be the planned one
                raise StitchingError("the pyrDown mode changed between plan() and run(): sharded multi-band blending needs the scalar order")
            blender = make_shard_blender(self.ctx, self.roi, self.req_bands)
            blender.set_band(*p.band(self.rank))
            send_msgs = p.sends(self.rank)
            recv_msgs = p.recvs(self.rank)
            # 1. the images that owe strips to other ranks: warp, feed, export, start the exchange
            senders = sorted({m[0] for m in send_msgs}) if self.split_boundary else list(self.my_orders)
            warped = self._warp_and_feed(blender, [k for k in self.my_orders if k in senders], p)
            sends = []
            if p.exchange == "strips":
                packed = strip_pack_batch(self.ctx, [(warped[k][0], warped[k][1], rect[0], rect[1]) for (k, _s, _d, rect, _n) in send_msgs],
                                          p.strip_flags)
                sends = [(dst, buf, nbytes) for (_k, _src, dst, _rect, nbytes), buf in zip(send_msgs, packed)]
            else:
                exported = blender.export_contribs([(k, p.band(dst)) for (k, _src, dst, _rect, _nbytes) in send_msgs])
                for (k, _src, dst, rect, nbytes), (packed, r) in zip(send_msgs, exported):
                    if r != rect:
                        raise StitchingError("contribution geometry differs from the plan")
                    sends.append((dst, packed, nbytes))
            del warped
            self.transport.start(sends, [(m[1], m[4]) for m in recv_msgs], self.ctx)
            # 2. the other images of this rank are warped and fed while the strips travel
            self._warp_and_feed(blender, [k for k in self.my_orders if k not in senders], p)
            if self.split_boundary:
                blender.build()  # ... and so are their pyramids, before this stream starts waiting for the exchange
            # (not split: another panorama in flight covers the exchange; everything is built by ONE set of launches in blend())
            # 3. received strips join the image table in global feed order; blend this rank's band
            rbufs = self.transport.finish(self.ctx)
            # every strip of this job comes from a u8 warp with a 0 / 255 mask (warp_images_and_masks on all ranks)
            if p.exchange == "strips":
                blender.feed_strips([(buf, m[3][2], m[3][3], (p.corners[m[0]][0] + m[3][0], p.corners[m[0]][1]), m[0])
                                     for m, buf in zip(recv_msgs, rbufs)], self._bin_flag | p.strip_flags)
            else:
                for m, buf in zip(recv_msgs, rbufs):
                    blender.feed_contrib(m[0], m[3], buf, self._bin_flag)
            pano, mask = blender.blend()
        finally:
            config.set_device_resident(prev)
        return pano, mask

    def _run_flat(self, p):
        """feather / "no": warp everything, send every other band its columns (+ halo), feed this band's blender its own columns and the
        received strips in global feed order, blend, crop the halo off."""
        imgs, masks, rois = self.warper.warp_images_and_masks(self.frames, self.cameras, compensator=self._compensator_for(self.my_orders),
                                                              camera_arrays=self._cam_arrays)
        masks = self._seam_resized(self.my_orders, masks)
        warped = {}
        for k, img, mask, roi in zip(self.my_orders, imgs, masks, rois):
            if roi[0:2] != p.corners[k]:
                raise StitchingError("warp roi changed between plan() and run()")
            warped[k] = (img, mask)
        send_msgs, recv_msgs = p.sends(self.rank), p.recvs(self.rank)
        packed = strip_pack_batch(self.ctx, [(warped[k][0], warped[k][1], rect[0], rect[1]) for (k, _s, _d, rect, _n) in send_msgs], p.strip_flags)
        self.transport.start([(dst, buf, nbytes) for (_k, _src, dst, _rect, nbytes), buf in zip(send_msgs, packed)],
                             [(m[1], m[4]) for m in recv_msgs], self.ctx)
        band_roi, (c0, c1) = p.band_roi(self.rank)
        blender = ShardBlender(self.ctx, _lib.BLEND_FEATHER if p.kind == "feather" else _lib.BLEND_NO, 0, self.sharpness, band_roi)
        items = []  # (global feed index, image, mask, corner)
        for k in self.my_orders:
            x0, x1 = p.own_columns(k, self.rank)
            if x1 > x0:
                img, mask = warped[k]
                whole = x0 == 0 and x1 == p.sizes[k][0]
                items.append((k, img if whole else img[:, x0:x1], mask if whole else mask[:, x0:x1], (p.corners[k][0] + x0, p.corners[k][1])))
        rbufs = self.transport.finish(self.ctx)
        for m, buf in zip(recv_msgs, rbufs):
            simg, smask = strip_unpack(buf, m[3][2], m[3][3], self._bin_flag | p.strip_flags)
            items.append((m[0], simg, smask, (p.corners[m[0]][0] + m[3][0], p.corners[m[0]][1])))
        # the plain blender overwrites and the feather blender adds fp32 weights: both in the order of the reference's feed loop
        for k, img, mask, corner in sorted(items, key=lambda it: it[0]):
            blender.feed_ex(img, mask, corner, k)
        pano, mask = blender.blend()
        return pano[:, c0:c1], mask[:, c0:c1]

    def _compensator_for(self, orders):
        """the job's compensator restricted to the images `orders` (global indices), in that order; None without one"""
        if self.compensator is None or self.compensator.compensator_type == "no":
            return None
        key = tuple(orders)
        c = self._local_comp.get(key)
        if c is None:
            from .exposure_error_compensator import ExposureErrorCompensator

            if self.compensator.gains is None:
                raise StitchingError("ExposureErrorCompensator.set_gains(gains) must be called before a sharded job runs")
            c = ExposureErrorCompensator(self.compensator.compensator_type, estimator=object())
            c.set_gains([self.compensator.gains[k] for k in orders])
            self._local_comp[key] = c
        return c

    def _seam_resized(self, orders, masks):
        """SeamFinder.resize of the low-resolution seam masks of the images `orders` onto their warped masks (one batched launch); the
        warped masks themselves without seam masks"""
        if self.seam_masks is None:
            return masks
        from .seam_finder import SeamFinder

        for k in orders:
            if k not in self._seam_dev:
                m = self.seam_masks[k]
                self._seam_dev[k] = as_device(m if isinstance(m, DeviceImage) else np.asarray(m.get() if hasattr(m, "get") else m), self.ctx)
        return SeamFinder.resize_all([self._seam_dev[k] for k in orders], masks)

    def _warp_and_feed(self, blender, orders, p):
        """-> {order: (warped image, mask)} of the images fed"""
        if not orders:
            return {}
        local = {k: i for i, k in enumerate(self.my_orders)}
        frames = [self.frames[local[k]] for k in orders]
        cams = [self.cameras[local[k]] for k in orders]
        idx = [local[k] for k in orders]
        ka = (np.ascontiguousarray(self._cam_arrays[0][idx]), np.ascontiguousarray(self._cam_arrays[1][idx]))
        imgs, masks, rois = self.warper.warp_images_and_masks(frames, cams, compensator=self._compensator_for(orders), camera_arrays=ka)
        masks = self._seam_resized(orders, masks)
        for k, img, mask, roi in zip(orders, imgs, masks, rois):
            if roi[0:2] != p.corners[k]:
                raise StitchingError("warp roi changed between plan() and run()")
            self._feed_own(blender, p, k, img, mask)
        return {k: (img, mask) for k, img, mask in zip(orders, imgs, masks)}

    def _feed_own(self, blender, p, k, img, mask):
        feed_own_image(blender, p, self.rank, k, img, mask)

    def gather(self, pano, mask):
        """Assemble the full panorama on rank 0 (host side, outside any timed region)."""
        band, bmask = np.asarray(pano), np.asarray(mask)
        if self.world == 1 or self.dist is None:
            return band, bmask
        parts = self.dist.gather((band, bmask), 0)
        if self.rank != 0:
            return None, None
        return np.concatenate([p[0] for p in parts], axis=1), np.concatenate([p[1] for p in parts], axis=1)


class loopback_bootstrap:
    """`with loopback_bootstrap(enabled):` RCCL's bootstrap sockets (ncclGetUniqueId / ncclCommInitRank) stay on the loopback
    interface while the block runs — interface discovery has been seen to stall on boxes without a network — unless the
    caller already chose an interface.  The previous environment is restored on exit, so nothing else in the process (a
    torch.distributed nccl group created later, a multi-node job) inherits the choice."""

    def __init__(self, enabled):
        self.enabled, self.prev = bool(enabled), None

    def __enter__(self):
        import os

        self.touched = self.enabled and "NCCL_SOCKET_IFNAME" not in os.environ
        if self.touched:
            os.environ["NCCL_SOCKET_IFNAME"] = "lo"
        return self

    def __exit__(self, *exc):
        import os

        if self.touched:
            os.environ.pop("NCCL_SOCKET_IFNAME", None)
        return False


def all_ranks_on_one_host(group):
    import socket

    return len(set(group.all_gather(socket.gethostname()))) == 1


def call_with_timeout(fn, seconds, what):
    """fn() on a helper thread; StitchingError when it has not returned after `seconds`.  For calls into librccl's bootstrap
    (ncclGetUniqueId, ncclCommInitRank), which on a node without a usable interface do not fail but wait forever: the job then
    falls back to the host-staged transport instead of hanging (the helper thread is a daemon and is left behind)."""
    import threading

    box = {}

    def run():
        try:
            box["v"] = fn()
        except BaseException as e:  # noqa: BLE001 - re-raised on the calling thread
            box["e"] = e

    t = threading.Thread(target=run, daemon=True, name="stx-rccl-bootstrap")
    t.start()
    t.join(seconds)
    if t.is_alive():
        raise StitchingError(f"{what} has not returned after {seconds:.0f} s (STITCHING_AMD_RCCL_TIMEOUT)")
    if "e" in box:
        raise box["e"]
    return box.get("v")


def rccl_timeout():
    return float(os.environ.get("STITCHING_AMD_RCCL_TIMEOUT", "120"))


def default_transport(ctx, rank, world, group):
    """RCCL when it initialises on this node, else the host-staged transport over the control-plane group."""
    if world == 1:
        return _NullTransport()
    if group is None:
        raise StitchingError("a control-plane group (stitching_amd.rendezvous.TcpGroup) is needed for more than one rank")
    import sys

    want = os.environ.get("STITCHING_AMD_TRANSPORT", "rccl")
    if want not in ("rccl", "host", "gloo"):  # "gloo": the name this switch had while the control plane was torch.distributed
        raise StitchingError(f"STITCHING_AMD_TRANSPORT={want!r} (rccl | host)")
    # Every rank issues the same sequence of collectives whatever fails where: (1) rank 0 ALWAYS broadcasts — the unique
    # id, or None when librccl is missing / refused or RCCL is not wanted; (2) ranks that got an id try to join the
    # communicator; (3) one all-reduce(MIN) decides for everybody.
    one_host = all_ranks_on_one_host(group)  # a collective: every rank calls it
    uid = None
    if rank == 0 and want == "rccl":
        try:
            with loopback_bootstrap(one_host):
                uid = call_with_timeout(RcclTransport.unique_id, rccl_timeout(), "ncclGetUniqueId")
        except Exception as e:  # noqa: BLE001
            print(f"[stitching_amd] rank 0: no RCCL unique id ({e}); using the host-staged transport", file=sys.stderr)
    uid = group.broadcast(uid, 0)
    ok, tr = 0, None
    if uid is not None:
        try:
            with loopback_bootstrap(one_host):
                tr = call_with_timeout(lambda: RcclTransport(ctx, rank, world, uid), rccl_timeout(), "ncclCommInitRank")
            ok = 1
        except Exception as e:  # noqa: BLE001 - any failure -> agree on the fallback below
            print(f"[stitching_amd] rank {rank}: RCCL transport unavailable ({e}); using the host-staged transport", file=sys.stderr)
    if group.all_reduce_min(ok) == 1:
        # every rank holds a communicator: before a panorama depends on it, one small exchange around the ring through the very calls
        # the job will make (start / finish on a context), checked byte for byte and bounded by the same watchdog — a transport that
        # initialises but does not deliver (first contact with a node's RCCL / xGMI set-up) must cost a message, not the job
        try:
            call_with_timeout(lambda: ring_probe(tr, rank, world), rccl_timeout(), "the RCCL ring probe")
            ok = 1
        except Exception as e:  # noqa: BLE001
            ok = 0
            print(f"[stitching_amd] rank {rank}: RCCL initialised but the ring probe failed ({e}); using the host-staged transport", file=sys.stderr)
        if group.all_reduce_min(ok) == 1:
            return tr
    if tr is not None:
        tr.close()
    return HostStagedTransport(group, ctx)


def ring_probe(transport, rank, world, nbytes=4096):
    """`nbytes` of a rank-specific pattern to rank + 1, the pattern of rank - 1 back, through transport.start / finish on a context of
    its own (a probe that hangs must not leave the job's stream waiting for it).  Raises StitchingError when the bytes differ."""
    from .device import Context

    def pattern(r):
        return ((np.arange(nbytes, dtype=np.uint32) * 2654435761 + 40503 * (r + 1)) >> 13).astype(np.uint8).reshape(1, nbytes)

    pctx = Context(transport.ctx.device)
    try:
        out = DeviceImage.from_numpy(pattern(rank), pctx)
        transport.start([((rank + 1) % world, out, nbytes)], [((rank - 1) % world, nbytes)], pctx)
        got = transport.finish(pctx)
        pctx.sync()
        if not np.array_equal(np.asarray(got[0]).reshape(-1)[:nbytes], pattern((rank - 1) % world).reshape(-1)):
            raise StitchingError(f"rank {rank}: the strip received from rank {(rank - 1) % world} is not the one it sent")
        del got, out
    finally:
        pctx.close()


class _NullTransport:
    name = "none"

    def exchange(self, sends, recvs):
        if sends or recvs:
            raise StitchingError("no transport for a single-rank job")
        return []

    def start(self, sends, recvs, ctx=None):
        self.exchange(sends, recvs)

    def finish(self, ctx=None):
        return []


def feed_own_image(blender, plan, rank, k, img, mask):
    """An image of this rank joins its blender.  exchange="strips": only the columns its own band depends on — a view,
    no copy — so that the pyramids of a wide image (the pitched rows of a multi-row panorama reach over several bands) are
    not built where only other ranks look; they build those parts from the strips they receive."""
    if plan.exchange != "strips":
        blender.feed_ex(img, mask, plan.corners[k], k)
        return
    (x0, x1), _ = blender.strip_rect(plan.sizes[k], plan.corners[k], plan.band(rank))
    if x1 <= x0:
        return  # nothing of this image reaches the rank's own band
    if x0 == 0 and x1 == plan.sizes[k][0]:
        blender.feed_ex(img, mask, plan.corners[k], k)
    else:
        blender.feed_ex(img[:, x0:x1], mask[:, x0:x1], (plan.corners[k][0] + x0, plan.corners[k][1]), k)


def _mask_is_binary(ctx, mask):
    fl = C.c_int()
    _lib.check(ctx._lib.stx_buf_flags(mask._h, C.byref(fl)))
    return bool(fl.value & _lib.CONTRIB_U8_BINARY)


# ------------------------------------------------------------------------------------ test helper
def virtual_sharded_flat_blend(ctx, warped, masks, corners, sizes, world, kind, sharpness=0.0, mask_bits=False):
    """virtual_sharded_blend for the feather (kind="feather", sharpness) and the plain (kind="no") blender"""
    n = len(warped)
    owners = owners_contiguous(n, world)
    d_imgs = [as_device(w, ctx) for w in warped]
    d_masks = [as_device(m, ctx) for m in masks]
    binary = [_mask_is_binary(ctx, m) for m in d_masks]
    plan = ShardPlan(corners, sizes, owners, world, None, "strips", mask_bits and all(binary), kind=kind, halo=feather_halo(sharpness))
    bands = []
    for g in range(world):
        band_roi, (c0, c1) = plan.band_roi(g)
        b = ShardBlender(ctx, _lib.BLEND_FEATHER if kind == "feather" else _lib.BLEND_NO, 0, sharpness, band_roi)
        strips = {m[0]: m for m in plan.recvs(g)}
        for k in range(n):
            if owners[k] == g:
                x0, x1 = plan.own_columns(k, g)
                if x1 > x0:
                    b.feed_ex(d_imgs[k][:, x0:x1], d_masks[k][:, x0:x1], (plan.corners[k][0] + x0, plan.corners[k][1]), k)
            elif k in strips:
                rect, nbytes = strips[k][3], strips[k][4]
                packed = strip_pack(ctx, d_imgs[k], d_masks[k], rect[0], rect[1], plan.strip_flags)
                assert packed.width * packed.height == nbytes
                simg, smask = strip_unpack(packed, rect[2], rect[3], (_lib.CONTRIB_U8_BINARY if binary[k] else 0) | plan.strip_flags)
                b.feed_ex(simg, smask, (plan.corners[k][0] + rect[0], plan.corners[k][1]), k)
        pano, mask = b.blend()
        bands.append((np.asarray(pano)[:, c0:c1], np.asarray(mask)[:, c0:c1]))
    return np.concatenate([p for p, _ in bands], axis=1), np.concatenate([m for _, m in bands], axis=1), plan


def virtual_sharded_blend(ctx, warped, masks, corners, sizes, world, num_bands, exchange="strips", mask_bits=False, balance="midway"):
    """All `world` ranks simulated in ONE process on one GPU: same kernels, same geometry, the
    exchange is a pointer hand-over.  Returns (panorama, mask, plan) as numpy arrays."""
    n = len(warped)
    owners = owners_contiguous(n, world)
    roi = Blender.result_roi(corners, sizes)
    probe = make_shard_blender(ctx, roi, num_bands)
    d_imgs = [as_device(w, ctx) for w in warped]
    d_masks = [as_device(m, ctx) for m in masks]
    binary = [_mask_is_binary(ctx, m) for m in d_masks]
    plan = ShardPlan(corners, sizes, owners, world, probe, exchange, mask_bits and all(binary), balance=balance)  # bits need 0 / 255 masks
    blenders = []
    for g in range(world):
        b = make_shard_blender(ctx, roi, num_bands)
        b.set_band(*plan.band(g))
        blenders.append(b)
    for k in range(n):
        feed_own_image(blenders[owners[k]], plan, owners[k], k, d_imgs[k], d_masks[k])
    for (k, src, dst, rect, nbytes) in plan.messages:
        if exchange == "strips":
            packed = strip_pack(ctx, d_imgs[k], d_masks[k], rect[0], rect[1], plan.strip_flags)
            assert packed.width * packed.height == nbytes
            simg, smask = strip_unpack(packed, rect[2], rect[3], (_lib.CONTRIB_U8_BINARY if binary[k] else 0) | plan.strip_flags)
            blenders[dst].feed_ex(simg, smask, (plan.corners[k][0] + rect[0], plan.corners[k][1]), k)
            continue
        packed, r = blenders[src].export_contrib(k, plan.band(dst))
        assert r == rect, (r, rect)
        fl = C.c_int()
        _lib.check(ctx._lib.stx_buf_flags(packed._h, C.byref(fl)))  # in-process hand-over: the flag rides along
        blenders[dst].feed_contrib(k, rect, packed, fl.value)
    bands = [b.blend() for b in blenders]
    pano = np.concatenate([np.asarray(p) for p, _ in bands], axis=1)
    mask = np.concatenate([np.asarray(m) for _, m in bands], axis=1)
    return pano, mask, plan
