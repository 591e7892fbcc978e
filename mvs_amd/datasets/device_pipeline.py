"""Input pipeline with the pixel work on the GPU.

The reference's loader (MVSNet/datasets/dtu_yao_eval.py:60-108) decodes five 1600x1200 JPEGs per
sample on the host, converts them to float32, divides, crops, transposes and stacks -- 113 MB of
float pixels per depth map through host memory and over PCIe, 5 decodes per depth map although a
scan has only 49 distinct images.  Here, per scan:

  * every image of the scan is decoded ONCE (a thread pool; PIL releases the GIL in its decoder)
    straight into one pinned uint8 buffer [n,1200,1600,3] -- a quarter of the float bytes;
  * that buffer and the parsed cam files (K, E) go up on a side stream;
  * mvs_images_u8_to_planar_f32 and mvs_proj_matrices_f32 produce the scan-resident tensors
    [n,3,1184,1600] float32 (1.1 GB for a DTU scan: nothing against 288 GB of HBM) and [n,4,4];
  * a sample is then a device-side gather of its V views; the next scan is prepared on a worker
    thread and the side stream while the current one is being swept.

The tensors a sample yields are bit-identical to the reference loader's (tests/test_io_golden.py)."""
import os
import threading
import warnings
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .. import ops
from .dtu_eval import read_cam_file, read_pair_file


class _Sample(dict):
    """A sample of DeviceScanPipeline: like the reference loader's dict, with the [1,V,3,H,W] image tensor gathered from the
    scan's resident images the first time it is asked for."""

    def __missing__(self, key):
        if key != "imgs":
            raise KeyError(key)
        v = self["scan_imgs"].index_select(0, self["view_slots"]).unsqueeze(0)
        self[key] = v
        return v

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


class DeviceScanPipeline:
    def __init__(self, datapath, listfile, nviews, ndepths=192, interval_scale=1.06, device="cuda:0",
                 decode_workers=None, image_hw=(1200, 1600), crop_bottom=16, intrinsics_div=4.0):
        self.datapath, self.nviews, self.ndepths, self.interval_scale = datapath, nviews, ndepths, interval_scale
        self.dev = torch.device(device)
        self.image_hw, self.crop_bottom, self.div = tuple(image_hw), crop_bottom, intrinsics_div
        with open(listfile) as f:
            self.scans = [ln.rstrip() for ln in f.readlines() if ln.strip()]
        self.metas = {s: read_pair_file(os.path.join(datapath, s, "pair.txt")) for s in self.scans}
        # few decoder threads: a scan's 49 JPEGs take ~0.15 s on four, a scan's sweep ~0.37 s; every further thread
        # only competes with the kernel-launching thread for the interpreter lock (scripts/pipe_probe.py)
        self.pool = ThreadPoolExecutor(max_workers=decode_workers or 4)
        self.side = torch.cuda.Stream(device=self.dev)
        self.stats = {"decoded": 0, "scans": 0}

    def __len__(self):
        return sum(len(m) for m in self.metas.values())

    # ---- one scan: decode once, upload uint8, normalise on the device
    def _decode_into(self, path, dst):
        from PIL import Image
        with Image.open(path) as im:
            a = np.asarray(im.convert("RGB") if im.mode != "RGB" else im)
        assert a.shape[:2] == self.image_hw, f"{path}: {a.shape[:2]} != {self.image_hw}"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")              # `a` wraps PIL's read-only bytes; it is only read
            src = torch.from_numpy(a)
        dst.copy_(src)                # decoder buffer -> pinned staging buffer: a torch copy, outside the interpreter lock

    def _prepare(self, scan):
        metas = self.metas[scan]
        views = sorted({v for ref, src in metas for v in [ref] + src[:self.nviews - 1]})
        slot = {v: i for i, v in enumerate(views)}
        Hs, Ws = self.image_hw
        pinned = torch.empty((len(views), Hs, Ws, 3), dtype=torch.uint8).pin_memory()
        futs = [self.pool.submit(self._decode_into, os.path.join(self.datapath, scan, "images", f"{v:0>8}.jpg"),
                                 pinned[i]) for v, i in slot.items()]
        cams = [read_cam_file(os.path.join(self.datapath, scan, "cams", f"{v:0>8}_cam.txt"), self.interval_scale,
                              intrinsics_div=1.0) for v in views]     # the division happens on the device
        K = torch.from_numpy(np.stack([c[0] for c in cams])).pin_memory()
        E = torch.from_numpy(np.stack([c[1] for c in cams])).pin_memory()
        # view slots of every sample of the scan, uploaded once: a per-sample torch.tensor(..., device=...) is a
        # pageable host-to-device copy, which blocks the launching thread until the stream has drained -- the
        # GPU then idles while the host prepares the next sample
        idx_all = torch.tensor([[slot[v] for v in [ref] + src[:self.nviews - 1]] for ref, src in metas],
                               dtype=torch.int64).pin_memory()
        for f in futs:
            f.result()
        self.stats["decoded"] += len(views)
        with torch.cuda.device(self.dev), torch.cuda.stream(self.side):
            u8 = pinned.to(self.dev, non_blocking=True)
            imgs = ops.images_u8_to_planar(u8, Hs - self.crop_bottom, Ws)
            proj = ops.proj_matrices(K.to(self.dev, non_blocking=True), E.to(self.dev, non_blocking=True), self.div)
            idx_dev = idx_all.to(self.dev, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.side)
        self.stats["scans"] += 1
        return {"scan": scan, "slot": slot, "imgs": imgs, "proj": proj, "done": done, "pinned": pinned,
                "idx": idx_dev, "idx_host": idx_all,
                "depth": {v: (c[2], c[3]) for v, c in zip(views, cams)}}

    def __iter__(self):
        nxt, box = None, {}

        def worker(scan):
            try:
                box["v"] = self._prepare(scan)
            except BaseException as e:   # surfaced on the consuming thread
                box["e"] = e

        for si, scan in enumerate(self.scans):
            if nxt is None:
                cur = self._prepare(scan)
            else:
                nxt.join()
                if "e" in box:
                    raise box.pop("e")
                cur = box.pop("v")
            nxt = None
            if si + 1 < len(self.scans):
                nxt = threading.Thread(target=worker, args=(self.scans[si + 1],), daemon=True)
                nxt.start()
            torch.cuda.current_stream(self.dev).wait_event(cur["done"])
            cur["imgs"].record_stream(torch.cuda.current_stream(self.dev))
            cur["proj"].record_stream(torch.cuda.current_stream(self.dev))
            cur["idx"].record_stream(torch.cuda.current_stream(self.dev))
            dv_cache = {}
            for k, (ref, src) in enumerate(self.metas[scan]):
                idx = cur["idx"][k]
                dmin, dint = cur["depth"][ref]
                if (dmin, dint) not in dv_cache:
                    dv_cache[(dmin, dint)] = torch.from_numpy(
                        np.arange(dmin, dint * (self.ndepths - 0.5) + dmin, dint, dtype=np.float32)).pin_memory().to(
                            self.dev, non_blocking=True)
                # "scan_imgs" / "view_slots": the scan's resident images and this sample's rows of them, for a
                # driver that runs FeatureNet once per image (MVSNet.extract_features)
                # "imgs" is gathered on first access (a driver that hands forward() cached feature maps never
                # reads it -- 113 MB of index_select per depth map otherwise)
                yield _Sample({"scan_imgs": cur["imgs"], "view_slots": idx,
                       "proj_matrices": cur["proj"].index_select(0, idx).unsqueeze(0),
                       "depth_values": dv_cache[(dmin, dint)].unsqueeze(0),
                       "filename": [scan + "/{}/" + f"{ref:0>8}" + "{}"]})
