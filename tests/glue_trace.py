"""Record / replay of everything the reference's glue asks of its back-end classes (TEST INFRASTRUCTURE, round 6).

The reference's composition code (`/root/reference/stitching/stitcher.py`, `cropper.py`, ...) cannot travel to the GPU box and there is
no GPU where it lives, so "run the reference's own Python over the product" is split in two:

  * here (reference present): the UNMODIFIED `stitching.Stitcher(...).stitch(images)` runs over recording proxies of its own classes
    (Warper, Blender, ExposureErrorCompensator, SeamFinder, Timelapser, Images — cv2 = tests/fake_cv2_glue.py, the oracle).  Every
    construction, method call, attribute read, generator step and static call the GLUE makes on them — not what the classes do
    inside — becomes one event: the arguments by provenance (frame i, camera i, result r of an earlier event, rows / columns of it:
    the cropper's slices), the result by shape, dtype and SHA-256.  That list is DATA: `tests/golden/reference_glue/*.json`,
    written by tools/make_reference_glue_golden.py.
  * on the GPU box: `Replayer` makes the same calls in the same order on the product's classes (lazy generators stepped when the
    glue stepped them, device images sliced where the cropper sliced) and compares every result with the recorded digest.

A value is one of: None / bool / int / float / str (themselves), {"tuple": [...]}, {"list": [...]}, {"cam": i}, {"frame": i},
{"ref": name} (+ "shape", "dtype", "sha" when it is a result), {"view": name, "y": [a, b], "x": [a, b]}, {"umat": value},
{"nd": nested list, "dtype": str} (small arrays by value), {"np": dtype, "v": x} (a numpy scalar: its type decides numpy's arithmetic),
{"enum": name}, {"gen": name}, {"obj": name}."""
import hashlib
import inspect
import json
from enum import Enum

import numpy as np

SMALL = 64  # arrays up to this many elements travel by value


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def _is_umat(v):
    return hasattr(v, "get") and not isinstance(v, (np.ndarray, dict)) and type(v).__name__ == "UMat"


class _Gen:
    """a generator the back end returned, stepped by the glue"""

    def __init__(self, rec, name, it):
        self._rec, self._name, self._it = rec, name, it

    def __iter__(self):
        return self

    def __next__(self):
        return self._rec._step(self)


class Recorder:
    def __init__(self, frames, cameras):
        self.events = []
        self.depth = 0
        self.frames, self.cameras = list(frames), list(cameras)
        self._known = {}  # id(object) -> (value token, the object: kept alive so that ids stay unique)
        self._n = 0
        self._pending_io = []
        for i, f in enumerate(self.frames):
            self._known[id(f)] = ({"frame": i}, f)

    # ------------------------------------------------------------------ values
    def _name(self, prefix):
        self._n += 1
        return f"{prefix}{self._n}"

    def _root(self, a):
        b = a
        while b is not None:
            hit = self._known.get(id(b))
            if hit is not None and isinstance(b, np.ndarray):
                return hit[0], b
            b = getattr(b, "base", None)
        return None, None

    def enc(self, v):
        """an ARGUMENT: by provenance"""
        if isinstance(v, np.floating):  # (numpy.float64 IS a Python float: test it first)
            pass
        elif v is None or isinstance(v, (bool, int, float, str)):
            return v
        if isinstance(v, (np.integer, np.bool_)):
            return int(v)
        if isinstance(v, np.floating):
            # the TYPE travels: `K[0, 0] *= aspect` (warper.py:90-93) is a float64 product for a numpy.float64 aspect — what
            # Images.get_ratio returns — and a float32 one for a Python float under NumPy 2's promotion rules
            return {"np": str(v.dtype), "v": float(v)}
        if isinstance(v, Enum):
            return {"enum": v.name}
        hit = self._known.get(id(v))
        if hit is not None:
            return dict(hit[0])
        if _is_umat(v):
            return {"umat": self.enc(v.get())}
        if isinstance(v, (list, tuple)):
            return {"tuple" if isinstance(v, tuple) else "list": [self.enc(x) for x in v]}
        if hasattr(v, "focal") and hasattr(v, "R"):
            for i, c in enumerate(self.cameras):
                if float(c.focal) == float(v.focal) and np.array_equal(np.asarray(c.R, np.float32), np.asarray(v.R, np.float32)):
                    return {"cam": i}
            raise ValueError("a camera that is none of the scenario's")
        if isinstance(v, np.ndarray):
            tok, root = self._root(v)
            if root is not None and root is not v:
                off = v.__array_interface__["data"][0] - root.__array_interface__["data"][0]
                s0, s1 = root.strides[0], root.strides[1]
                y0, x0 = off // s0, (off % s0) // s1
                assert off == y0 * s0 + x0 * s1 and v.strides[:2] == root.strides[:2] and v.shape[2:] == root.shape[2:], "not a rectangle of its base"
                return {"view": tok.get("ref", tok), "y": [int(y0), int(y0 + v.shape[0])], "x": [int(x0), int(x0 + v.shape[1])]}
            if v.size <= SMALL:
                return {"nd": v.tolist(), "dtype": str(v.dtype)}
            raise ValueError(f"an array argument of unknown provenance, shape {v.shape}: the glue made it itself")
        raise ValueError(f"cannot record an argument of type {type(v)}")

    def enc_ret(self, v):
        """a RESULT: by content; arrays, generators and instances get a name later arguments refer to"""
        if isinstance(v, np.floating):  # (numpy.float64 IS a Python float: test it first)
            pass
        elif v is None or isinstance(v, (bool, int, float, str)):
            return v
        if isinstance(v, (np.integer, np.bool_)):
            return int(v)
        if isinstance(v, np.floating):
            return {"np": str(v.dtype), "v": float(v)}
        if isinstance(v, Enum):
            return {"enum": v.name}
        if isinstance(v, _Gen):
            return {"gen": v._name}
        if _is_umat(v):
            return {"umat": self.enc_ret(v.get())}
        if isinstance(v, (list, tuple)):
            return {"tuple" if isinstance(v, tuple) else "list": [self.enc_ret(x) for x in v]}
        if isinstance(v, np.ndarray):
            hit = self._known.get(id(v))
            if hit is not None and "ref" not in hit[0]:  # an input handed back unchanged (e.g. the "no" compensator)
                tok = dict(hit[0])
            else:
                name = hit[0]["ref"] if hit is not None else self._name("r")
                self._known[id(v)] = ({"ref": name}, v)
                tok = {"ref": name}
            tok.update(shape=list(v.shape), dtype=str(v.dtype), sha=sha(v))
            return tok
        hit = self._known.get(id(v))
        if hit is not None:
            return dict(hit[0])
        raise ValueError(f"cannot record a result of type {type(v)}")

    # ------------------------------------------------------------------ events
    def _log(self, ev):
        self.events.append(ev)
        self.events.extend(self._pending_io)  # images written while the call ran follow it
        del self._pending_io[:]

    def _call(self, ev, fn, a, kw):
        """one call of the glue into the back end; calls the back end makes itself while it runs are not events"""
        top = self.depth == 0
        if top:
            ev["args"] = [self.enc(x) for x in a]
            if kw:
                ev["kwargs"] = {k: self.enc(x) for k, x in kw.items()}
        self.depth += 1
        try:
            ret = fn(*a, **kw)
        except Exception as e:
            if top:
                ev["raises"] = type(e).__name__
                self._log(ev)
            raise
        finally:
            self.depth -= 1
        if inspect.isgenerator(ret):
            ret = _Gen(self, self._name("g"), ret)
            self._known[id(ret)] = ({"gen": ret._name}, ret)
        if top:
            if ev["op"] != "new":
                ev["ret"] = self.enc_ret(ret)
            self._log(ev)
        return ret

    def _step(self, gen):
        top = self.depth == 0
        self.depth += 1
        try:
            v = next(gen._it)
        except StopIteration:
            if top:
                self._log({"op": "next", "gen": gen._name, "stop": True})
            raise
        finally:
            self.depth -= 1
        if top:
            self._log({"op": "next", "gen": gen._name, "ret": self.enc_ret(v)})
        return v

    def io(self, name, arr):
        """an image the back end wrote through cv.imwrite (the timelapser's frames): observed, attached to the running call.  What the
        GLUE writes itself (verbose.py's write_verbose_result) is not the back end's doing and is not recorded."""
        if self.depth == 0:
            return
        self._pending_io.append({"op": "io", "name": name, "shape": list(arr.shape), "dtype": str(arr.dtype), "sha": sha(arr)})

    # ------------------------------------------------------------------ proxies
    def proxy_class(self, real, label):
        """a class that stands where the glue names `real`: instances delegate to a `real` instance, everything the glue does is logged"""
        rec = self

        class Proxy:
            _real_cls = real

            def __init__(self, *a, **kw):
                name = rec._name("o")
                object.__setattr__(self, "_rec_name", name)
                rec._known[id(self)] = ({"obj": name}, self)
                object.__setattr__(self, "_real", rec._call({"op": "new", "cls": label, "obj": name}, real, a, kw))

            def __getattr__(self, attr):
                val = getattr(object.__getattribute__(self, "_real"), attr)
                name = object.__getattribute__(self, "_rec_name")
                if callable(val) and not isinstance(val, type):
                    return lambda *a, **kw: rec._call({"op": "call", "obj": name, "name": attr}, val, a, kw)
                if rec.depth == 0 and not attr.startswith("_"):
                    rec._log({"op": "get", "obj": name, "name": attr, "ret": rec.enc_ret(val) if not isinstance(val, (dict, set)) else None})
                return val

            def __setattr__(self, attr, val):
                setattr(object.__getattribute__(self, "_real"), attr, val)

        def adopt(inst):
            """an instance a static factory returned (Images.of): wrap it"""
            p = object.__new__(Proxy)
            name = rec._name("o")
            object.__setattr__(p, "_rec_name", name)
            object.__setattr__(p, "_real", inst)
            rec._known[id(p)] = ({"obj": name}, p)
            return p

        for attr in dir(real):
            if attr.startswith("__"):
                continue
            raw = inspect.getattr_static(real, attr)
            val = getattr(real, attr)
            if isinstance(raw, (staticmethod, classmethod)):
                def make(attr=attr, val=val):
                    def static(*a, **kw):
                        def run(*a2, **kw2):
                            r = val(*a2, **kw2)
                            return adopt(r) if isinstance(r, real) else r
                        return rec._call({"op": "static", "cls": label, "name": attr}, run, a, kw)
                    return static
                setattr(Proxy, attr, staticmethod(make()))
            elif not callable(val) or isinstance(val, type):
                setattr(Proxy, attr, val)  # constants, nested enums
        Proxy.__name__ = Proxy.__qualname__ = real.__name__
        return Proxy

    def dump(self, path, meta):
        with open(path, "w") as f:
            json.dump({"meta": meta, "events": self.events}, f, indent=0, separators=(",", ":"))
            f.write("\n")


class ReplayMismatch(AssertionError):
    pass


class _UMatLike:
    """what the reference hands on where cv2 produced a cv.UMat (seam masks): the array is only reachable through .get()"""

    def __init__(self, a):
        self._a = a

    def get(self):
        return self._a


# SeamFinder's plot helpers (verbose mode: seam_finder.py:45-75): cv2 drawing code of the reference's own, whatever is switched — the
# product hands them to the reference's class, which does not exist on the GPU box.  Never replayed; nothing later refers to their results.
PLOT_HELPERS = ("draw_seam_mask", "draw_seam_polygons", "draw_seam_lines", "extract_seam_lines")


class Replayer:
    """Makes the recorded calls on `classes` ({label: class}) and compares every result with its digest.  Events of a label that is
    NOT in `classes` are served by `fallback[label]` — a CPU stand-in of the reference's class (the oracle) for the partial switch of
    INTEGRATION.md §1 — and compared all the same.  `imwrite_log`: the list the test's cv2.imwrite stand-in appends (name, array) to."""

    def __init__(self, trace, classes, frames, cameras, fallback=None, imwrite_log=None, umat=_UMatLike, compare=True):
        self.events = trace["events"]
        self.classes, self.fallback = classes, fallback or {}
        self.frames, self.cameras = frames, cameras
        self.tab = {}
        self.obj_label = {}
        self.imwrite_log = imwrite_log if imwrite_log is not None else []
        self.umat = umat
        self.checked = 0
        # compare=False: the calls are made, the results kept (self.tab), no array is compared — for a back end switched to another
        # arithmetic model than the recording's (tests/test_gpu_opencv_golden.py compares the panorama with OpenCV's instead)
        self.compare = compare

    def _cls(self, label):
        return self.classes[label] if label in self.classes else self.fallback[label]

    def dec(self, v):
        if not isinstance(v, dict):
            return v
        if "tuple" in v:
            return tuple(self.dec(x) for x in v["tuple"])
        if "list" in v:
            return [self.dec(x) for x in v["list"]]
        if "cam" in v:
            return self.cameras[v["cam"]]
        if "frame" in v:
            return self.frames[v["frame"]]
        if "view" in v:
            base = self.tab[v["view"]] if not isinstance(v["view"], dict) else self.dec(v["view"])
            return base[v["y"][0]:v["y"][1], v["x"][0]:v["x"][1]]
        if "ref" in v:
            return self.tab[v["ref"]]
        if "gen" in v:
            return self.tab[v["gen"]]
        if "obj" in v:
            return self.tab[v["obj"]]
        if "umat" in v:
            x = self.dec(v["umat"])  # a named result comes back in the form its producer returned it (cv.UMat-like or plain)
            return x if isinstance(v["umat"], dict) and "ref" in v["umat"] else self.umat(x)
        if "nd" in v:
            return np.asarray(v["nd"], dtype=v["dtype"])
        if "np" in v:
            return np.dtype(v["np"]).type(v["v"])
        if "enum" in v:
            return self._cls("Images").Resolution[v["enum"]]
        raise ValueError(f"unknown value {v}")

    def check(self, exp, got, where):
        if isinstance(exp, dict) and "umat" in exp:
            inner = got.get() if hasattr(got, "get") and not isinstance(got, np.ndarray) else got
            self.check(exp["umat"], inner, where)
            # the live object goes into the table as the back end returned it (a cv.UMat from cv2's finders, an array from the product)
            if isinstance(exp["umat"], dict) and "ref" in exp["umat"]:
                self.tab[exp["umat"]["ref"]] = got
            return
        if isinstance(exp, dict) and ("tuple" in exp or "list" in exp):
            items = exp.get("tuple", exp.get("list"))
            got = list(got)
            if len(got) != len(items):
                raise ReplayMismatch(f"{where}: {len(got)} values, recorded {len(items)}")
            for i, (e, g) in enumerate(zip(items, got)):
                self.check(e, g, f"{where}[{i}]")
            return
        if isinstance(exp, dict) and "sha" in exp:
            a = np.asarray(got.get() if hasattr(got, "get") and not isinstance(got, np.ndarray) else got)
            if list(a.shape) != exp["shape"] or str(a.dtype) != exp["dtype"]:
                raise ReplayMismatch(f"{where}: {a.shape} {a.dtype}, recorded {exp['shape']} {exp['dtype']}")
            if self.compare and sha(a) != exp["sha"]:
                raise ReplayMismatch(f"{where}: contents differ from the recording ({a.shape} {a.dtype})")
            if "ref" in exp:
                self.tab[exp["ref"]] = got
            self.checked += 1
            return
        if isinstance(exp, dict) and "gen" in exp:
            if not hasattr(got, "__next__"):
                raise ReplayMismatch(f"{where}: the reference returns a generator here, got {type(got)}")
            self.tab[exp["gen"]] = got
            return
        if isinstance(exp, dict) and "obj" in exp:
            self.tab[exp["obj"]] = got
            return
        if isinstance(exp, dict) and "enum" in exp:
            if getattr(got, "name", None) != exp["enum"]:
                raise ReplayMismatch(f"{where}: {got!r}, recorded {exp['enum']}")
            return
        if isinstance(exp, dict) and "np" in exp:
            # value AND type: what the glue passes on (an aspect) must behave in numpy arithmetic as the reference's value does
            if not isinstance(got, np.generic) or str(got.dtype) != exp["np"] or float(got) != exp["v"]:
                raise ReplayMismatch(f"{where}: {got!r} ({type(got).__name__}), recorded numpy.{exp['np']}({exp['v']!r})")
            return
        if isinstance(exp, dict) and ("frame" in exp or "cam" in exp):
            if got is not self.dec(exp):
                raise ReplayMismatch(f"{where}: recorded the input itself ({exp})")
            return
        if isinstance(got, (np.integer, np.floating, np.bool_)):
            got = got.item()
        if isinstance(exp, float) or isinstance(got, float):
            if exp is None or got is None or float(exp) != float(got):
                raise ReplayMismatch(f"{where}: {got!r}, recorded {exp!r}")
            return
        if exp != got:
            raise ReplayMismatch(f"{where}: {got!r}, recorded {exp!r}")

    def run(self):
        i, n = 0, len(self.events)
        while i < n:
            ev = self.events[i]
            where = f"event {i} {ev.get('cls', ev.get('obj', ev.get('gen', '')))}.{ev.get('name', ev['op'])}"
            a = [self.dec(x) for x in ev.get("args", [])]
            kw = {k: self.dec(x) for k, x in ev.get("kwargs", {}).items()}
            io_before = len(self.imwrite_log)
            if ev.get("name") in PLOT_HELPERS:
                i += 1
                continue
            try:
                if ev["op"] == "new":
                    got = self._cls(ev["cls"])(*a, **kw)
                    self.tab[ev["obj"]] = got
                    self.obj_label[ev["obj"]] = ev["cls"]
                elif ev["op"] == "call":
                    got = getattr(self.tab[ev["obj"]], ev["name"])(*a, **kw)
                elif ev["op"] == "static":
                    got = getattr(self._cls(ev["cls"]), ev["name"])(*a, **kw)
                elif ev["op"] == "get":
                    got = getattr(self.tab[ev["obj"]], ev["name"])
                elif ev["op"] == "next":
                    try:
                        got = next(self.tab[ev["gen"]])
                    except StopIteration:
                        if not ev.get("stop"):
                            raise ReplayMismatch(f"{where}: exhausted, the recording yields a value")
                        i += 1
                        continue
                    if ev.get("stop"):
                        raise ReplayMismatch(f"{where}: yields a value, the recording is exhausted")
                elif ev["op"] == "io":
                    raise ReplayMismatch(f"{where}: an image was written in the recording without a call in front of it")
                else:
                    raise ValueError(ev["op"])
            except ReplayMismatch:
                raise
            except Exception as e:
                if ev.get("raises"):
                    i += 1
                    continue
                raise ReplayMismatch(f"{where}: raised {type(e).__name__}: {e}") from e
            if ev.get("raises"):
                raise ReplayMismatch(f"{where}: the reference raises {ev['raises']} here")
            if ev["op"] != "new" and not (ev["op"] == "get" and ev.get("ret") is None):
                self.check(ev.get("ret"), got, where)
            i += 1
            # images the call wrote (timelapse frames) follow it in the recording
            k = io_before
            while i < n and self.events[i]["op"] == "io":
                io = self.events[i]
                if k >= len(self.imwrite_log):
                    raise ReplayMismatch(f"{where}: the reference wrote {io['name']} here, nothing was written")
                name, arr = self.imwrite_log[k]
                arr = np.asarray(arr)
                if name != io["name"] or list(arr.shape) != io["shape"] or sha(arr) != io["sha"]:
                    raise ReplayMismatch(f"{where}: written image {name} {arr.shape} differs from the recording ({io['name']} {io['shape']})")
                self.checked += 1
                k += 1
                i += 1
        return self.checked


def load(path):
    with open(path) as f:
        return json.load(f)
