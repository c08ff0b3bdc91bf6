"""Seeding, latent-trajectory store and frame I/O with the reference's file formats
(``i2vgen-xl/utils.py:17-79``; writer ``pipeline_i2vgen_xl.py:1424-1428``)."""
from __future__ import annotations

import contextlib
import glob
import logging
import os
import random
import shutil
import itertools
import threading
from typing import Dict, List, Optional

import numpy as np
import torch
from PIL import Image

logger = logging.getLogger(__name__)


def seed_everything(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    random.seed(seed)
    np.random.seed(seed)


_PENDING: Dict[str, "LatentTrajectory"] = {}   # output_dir (absolute) -> trajectory whose writer thread is still running
_PENDING_LOCK = threading.Lock()


def wait_for_pending_writes(ddim_latents_path=None):
    """Join the background writer of ``ddim_latents_path`` (all writers when None).  Every reader of ``ddim_latents_*.pt``
    in this module calls it first, so ``pipe.invert(output_dir=d)`` followed by ``sample_with_pnp(ddim_inv_latents_path=d)``
    in one process -- the reference's demo flow, where each file is written synchronously inside the loop
    (``pipeline_i2vgen_xl.py:1424-1428``) -- never sees a missing or truncated file."""
    with _PENDING_LOCK:
        if ddim_latents_path is None:
            trajs = list(_PENDING.values())
        else:
            t = _PENDING.get(os.path.abspath(str(ddim_latents_path)))
            trajs = [t] if t is not None else []
    for tr in trajs:
        tr.wait()


def load_ddim_latents_at_t(t, ddim_latents_path):
    """``i2vgen-xl/utils.py:25-30``.  ``ddim_latents_path`` may also be an in-memory ``LatentTrajectory``."""
    if isinstance(ddim_latents_path, LatentTrajectory):
        return ddim_latents_path[int(t)]
    wait_for_pending_writes(ddim_latents_path)
    ddim_latents_at_t_path = os.path.join(ddim_latents_path, f"ddim_latents_{int(t)}.pt")
    assert os.path.exists(ddim_latents_at_t_path), f"Missing latents at t {t} path {ddim_latents_at_t_path}"
    ddim_latents_at_t = torch.load(ddim_latents_at_t_path, map_location="cpu")
    logger.debug(f"Loaded ddim_latents_at_t from {ddim_latents_at_t_path}")
    return ddim_latents_at_t


def load_ddim_latents_at_T(ddim_latents_path):
    if isinstance(ddim_latents_path, LatentTrajectory):
        return ddim_latents_path[max(ddim_latents_path.keys())]
    wait_for_pending_writes(ddim_latents_path)
    noisest = max(int(x.split("_")[-1].split(".")[0])
                  for x in glob.glob(os.path.join(ddim_latents_path, "ddim_latents_*.pt")))
    return torch.load(os.path.join(ddim_latents_path, f"ddim_latents_{noisest}.pt"), map_location="cpu")


def inversion_is_complete(output_dir, latents_dir=None) -> bool:
    """Stage-1 skip criterion.  The reference skips an entry when ``output_dir`` exists
    (``run_group_ddim_inversion.py:118-120``); here additionally no half-written latents directory (``*.partial-<pid>``,
    see ``LatentTrajectory.save``) may sit next to ``latents_dir`` -- a run that crashed mid-write is redone, not skipped."""
    if not os.path.exists(output_dir):
        return False
    return not (latents_dir and glob.glob(os.path.abspath(str(latents_dir)) + ".partial-*"))


class LatentTrajectory:
    """The inversion trajectory {t: latents[1,4,F,h,w]} kept resident in HBM (25 MB for 50 steps at 16f x 512^2)
    instead of the reference's blocking ``torch.save`` / ``torch.load`` round trip inside both hot loops.
    ``save`` writes the reference's on-disk format (``ddim_latents_{t}.pt``) from a background thread."""

    _serials = itertools.count(1)

    def __init__(self):
        self._lat: Dict[int, torch.Tensor] = {}
        self._writer: Optional[threading.Thread] = None
        self.serial = next(LatentTrajectory._serials)   # process-unique (``id()`` can be reused after the object is freed)

    def __setitem__(self, t, x):
        self._lat[int(t)] = x

    def __getitem__(self, t):
        t = int(t)
        assert t in self._lat, f"Missing latents at t {t}"
        return self._lat[t]

    def __contains__(self, t):
        return int(t) in self._lat

    def keys(self):
        return self._lat.keys()

    def __len__(self):
        return len(self._lat)

    def save(self, output_dir: str, background: bool = False):
        """Write ``ddim_latents_{t}.pt`` (the reference's format).  Every file is written under a temporary name and
        renamed, and a directory that did not exist before appears under its final name only when it is complete: the
        runners use "output_dir exists" as the stage-1 skip criterion (``run_group_ddim_inversion.py:118-120``), so a crash
        mid-write must not leave a partial directory behind.  ``background=True`` returns at once; readers in this module
        join the writer first (``wait_for_pending_writes``)."""
        final = os.path.abspath(output_dir)
        self.wait()
        for stale in glob.glob(final + ".partial-*"):   # left behind by a run that died mid-write
            shutil.rmtree(stale, ignore_errors=True)
        fresh = not os.path.exists(final)
        stage = f"{final}.partial-{os.getpid()}" if fresh else final
        os.makedirs(stage, exist_ok=True)
        host = {t: x.detach().to("cpu") for t, x in self._lat.items()}  # one sync for the whole trajectory

        def work():
            try:
                for t, x in host.items():
                    tmp = os.path.join(stage, f".ddim_latents_{t}.pt.tmp-{os.getpid()}")
                    torch.save(x, tmp)
                    os.replace(tmp, os.path.join(stage, f"ddim_latents_{t}.pt"))
                if fresh:
                    try:
                        os.rename(stage, final)
                    except OSError:  # somebody created `final` meanwhile: move the files over one by one
                        os.makedirs(final, exist_ok=True)
                        for name in os.listdir(stage):
                            os.replace(os.path.join(stage, name), os.path.join(final, name))
                        os.rmdir(stage)
            finally:
                with _PENDING_LOCK:
                    if _PENDING.get(final) is self:
                        del _PENDING[final]

        if background:
            with _PENDING_LOCK:
                _PENDING[final] = self
            self._writer = threading.Thread(target=work, daemon=False)
            self._writer.start()
        else:
            work()

    def wait(self):
        w = self._writer
        if w is not None and w is not threading.current_thread():
            w.join()
            self._writer = None

    @classmethod
    def load(cls, ddim_latents_path: str, device=None, timesteps=None) -> "LatentTrajectory":
        tr = cls()
        wait_for_pending_writes(ddim_latents_path)
        if timesteps is None:
            timesteps = [int(x.split("_")[-1].split(".")[0])
                         for x in glob.glob(os.path.join(ddim_latents_path, "ddim_latents_*.pt"))]
        for t in timesteps:
            x = load_ddim_latents_at_t(int(t), ddim_latents_path)
            tr[int(t)] = x.to(device) if device is not None else x
        return tr


def load_image(image) -> Image.Image:
    """[3P] ``diffusers.utils.load_image`` (called at ``run_group_pnp_edit.py:120``, ``gradio_demo.py:151``): a path or a PIL image ->
    EXIF orientation applied (a phone's JPEG of an edited first frame is stored rotated), RGB.  No URLs here (no network)."""
    from PIL import ImageOps
    if isinstance(image, (str, os.PathLike)):
        if str(image).startswith(("http://", "https://")):
            raise ValueError(f"load_image: {image} -- no network access here, give a local file")
        if not os.path.isfile(image):
            raise ValueError(f"Incorrect path or URL. URLs must start with `http://` or `https://`, and {image} is not a valid path.")
        image = Image.open(image)
    elif not isinstance(image, Image.Image):
        raise ValueError("Incorrect format used for the image. Should be a URL linking to an image, a local path, or a PIL image.")
    return ImageOps.exif_transpose(image).convert("RGB")


def load_video_frames(frames_path, n_frames, image_size=(512, 512)):
    """``i2vgen-xl/utils.py:70-79``."""
    paths = [f"{frames_path}/%05d.png" % i for i in range(n_frames)]
    frames = [load_image(p) for p in paths]
    for f in frames:
        if f.size != tuple(image_size):
            logger.error(f"Frame size {f.size} does not match config.image_size {image_size}")
            raise ValueError(f"Frame size {f.size} does not match config.image_size {image_size}")
    return paths, frames


def isinstance_str(x: object, cls_name: str):
    """``consisti2v/utils.py:13-25`` / ``seine/utils.py``: does ``x`` have a class NAMED ``cls_name`` in its ancestry."""
    return any(c.__name__ == cls_name for c in type(x).__mro__)


def convert_video_to_frames(video_path, img_size=(512, 512), save_frames=True, save_dir=None):
    """``i2vgen-xl/utils.py:43-67`` / ``consisti2v/utils.py:55-76`` (which adds ``save_dir``): decode the clip, LANCZOS-resize every frame
    to ``img_size``, optionally save them as ``<save_dir or video dir>/<video name>/%05d.png``.  The reference decodes with torchvision / ffmpeg; this image has no codec library,
    so only mp4 files whose H.264 pictures are raw (I_PCM) macroblocks -- what ``export_to_video`` below writes -- can be read
    (``anyv2v_amd.mp4``); anything else raises with the reason and the frame-directory alternative."""
    from .mp4 import Mp4Unsupported, read_mp4
    try:
        video, _fps = read_mp4(video_path)
    except Mp4Unsupported as e:
        raise RuntimeError(f"cannot decode {video_path}: {e}; no video decoder library is available here (no torchvision / "
                           "ffmpeg / cv2) -- provide the frames as a directory of %05d.png files") from e
    if save_frames:
        out_dir = os.path.join(str(save_dir) if save_dir is not None else os.path.dirname(video_path),
                               os.path.splitext(os.path.basename(video_path))[0])
        os.makedirs(out_dir, exist_ok=True)
    frames = []
    for i, image in enumerate(video):
        if image.size != tuple(img_size):
            image = image.resize(tuple(img_size), resample=Image.Resampling.LANCZOS)
        if save_frames:
            image.save(os.path.join(out_dir, f"{i:05d}.png"))
        frames.append(image)
    return frames


def export_to_video(frames: List[Image.Image], path: str, fps: int = 8):
    """``diffusers.utils.export_to_video`` as called at ``i2vgen-xl/run_group_pnp_edit.py:178``: an H.264 mp4 any player opens
    (raw I_PCM macroblocks: there is no encoder library here; see ``anyv2v_amd.mp4``)."""
    from .mp4 import write_mp4
    return write_mp4(frames, path, fps=fps)


def export_to_gif(frames: List[Image.Image], path: Optional[str] = None, fps: int = 10):
    """[3P] ``diffusers.utils.export_to_gif`` (0.26: a fixed 100 ms per frame, endless loop, no palette optimisation; later releases
    added ``fps`` with the same default) -- the reference's runners never pass a rate (``run_group_ddim_inversion.py:135,188``,
    ``run_group_pnp_edit.py:179``), so their GIFs play at 10 frames per second whatever ``target_fps`` says."""
    if path is None:
        import tempfile
        path = tempfile.NamedTemporaryFile(suffix=".gif", delete=False).name
    frames[0].save(path, save_all=True, append_images=frames[1:], optimize=False, duration=1000 // fps, loop=0)
    return path


_CAPTURE_LOCK = threading.RLock()
_CAPTURE_DEPTH = 0          # captures in progress in this process (nested, or on several threads: two step engines, the trajectory writer)
_CAPTURE_GC_WAS_ON = False  # the collector's state when the outermost capture began
_DEFERRED_GRAPHS: list = []  # graph objects released while a capture was recording: kept alive until no stream captures


def release_graphs(graphs) -> None:
    """Drop the ``CUDAGraph`` objects of an evicted step engine (``graphs``: a dict or list, emptied here).  Their destructor calls
    ``hipGraphExecDestroy``, which is not permitted while ANY stream of the process is capturing -- and a refcount-driven free is not
    the cyclic collector's doing, so switching the collector off does not cover it: while a capture is recording the objects are
    parked and freed when the last capture ends."""
    with _CAPTURE_LOCK:
        objs = list(graphs.values()) if isinstance(graphs, dict) else list(graphs)
        graphs.clear()
        if _CAPTURE_DEPTH > 0:
            _DEFERRED_GRAPHS.extend(objs)


@contextlib.contextmanager
def capture_hip_graph(graph):
    """``with torch.cuda.graph(graph)`` made safe against the cyclic garbage collector: an unreachable ``CUDAGraph`` (a step engine dropped
    by the LRU, a previous pipeline object) that Python's GC happens to free DURING a capture calls ``hipGraphExecDestroy`` on the capturing
    thread -- "operation not permitted when stream is capturing", and the process aborts in the destructor.  torch >= 2.9 no longer
    collects before a capture by default, so: collect first, keep the collector off while the stream captures.  Re-entrant and
    thread-safe: a depth counter under a lock, the collector comes back on only when the LAST capture has ended (an inner or concurrent
    capture must not re-enable it under an outer one), and graphs released meanwhile (``release_graphs``) are freed then."""
    global _CAPTURE_DEPTH, _CAPTURE_GC_WAS_ON
    import gc
    import torch
    with _CAPTURE_LOCK:
        if _CAPTURE_DEPTH == 0:
            gc.collect()
            _CAPTURE_GC_WAS_ON = gc.isenabled()
            gc.disable()
        _CAPTURE_DEPTH += 1
    try:
        with torch.cuda.graph(graph):
            yield
    finally:
        with _CAPTURE_LOCK:
            _CAPTURE_DEPTH -= 1
            if _CAPTURE_DEPTH == 0:
                parked = list(_DEFERRED_GRAPHS)
                _DEFERRED_GRAPHS.clear()
                del parked   # (freed here, outside every capture)
                if _CAPTURE_GC_WAS_ON:
                    gc.enable()
