#!/usr/bin/env python3
"""refrun.py — drive the oracle/_ref binaries (TEST INFRASTRUCTURE).

The reference has no CLI flag for bounces and writes its output relative to its cwd
(SURVEY.md §8(c)), so every run goes through a rewritten scene JSON piped on stdin with
cwd = the asset overlay `oracle/_ref/input/` (a copy of the reference's input/ plus generated
stand-in meshes). Used by tools/gen_golden.py, tests/ and bench.py's cpu_baseline leg.
"""
import json
import os
import subprocess
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(REPO, "oracle", "_ref")
INPUT_DIR = os.path.join(REF_DIR, "input")


def rewrite_scene(scene_name, width=None, height=None, samples=None, bounces=None, tile=None, out_dir=None,
                  file_type="bmp"):
    """Load input/<scene_name> from the overlay and override the renderer prefs."""
    with open(os.path.join(INPUT_DIR, scene_name)) as f:
        scene = json.load(f)
    r = scene["renderer"]
    if width is not None:
        r["width"] = int(width)
    if height is not None:
        r["height"] = int(height)
    if samples is not None:
        r["samples"] = int(samples)
    if bounces is not None:
        r["bounces"] = int(bounces)
    if tile is not None:
        r["tileWidth"], r["tileHeight"] = int(tile[0]), int(tile[1])
    r["outputFilePath"] = (out_dir.rstrip("/") + "/") if out_dir else "/tmp/"
    r["fileType"] = file_type
    r["outputFileName"] = "refrun"
    return scene


def _run(binary, scene_json, threads, env_extra, timeout, extra_args=()):
    env = dict(os.environ)
    env.update(env_extra)
    exe = os.path.join(REF_DIR, binary)
    if not os.path.exists(exe):
        raise FileNotFoundError(f"{exe} missing: run `make -C oracle ref` where /root/reference exists")
    args = [exe]
    if threads:
        args += ["-j", str(int(threads))]
    args += list(extra_args)
    proc = subprocess.run(args, input=json.dumps(scene_json).encode(), cwd=INPUT_DIR, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    if proc.returncode != 0:
        raise RuntimeError(f"{binary} failed ({proc.returncode}):\n{proc.stdout.decode(errors='replace')[-2000:]}")
    return proc.stdout.decode(errors="replace")


def render_reference(scene_name, width, height, samples, bounces, flavour="strict", threads=None, timeout=3600,
                     tile=None, iterative=False, env=None):
    """Run the real reference; returns (float32 array [H, W, 3] in the reference's stored row order, stats dict).
    iterative=True: `--iterative` (renderThreadInteractive, Halton sampler) on ONE thread — with more threads the
    reference races on state.finishedPasses and its output changes from run to run."""
    binary = {"default": "c-ray-ref", "strict": "c-ray-ref-strict", "count": "c-ray-ref-count"}[flavour]
    with tempfile.TemporaryDirectory() as tmp:
        scene = rewrite_scene(scene_name, width, height, samples, bounces, tile=tile, out_dir=tmp)
        f32 = os.path.join(tmp, "buffer.f32")
        stats = os.path.join(tmp, "stats.json")
        log = _run(binary, scene, 1 if iterative else (threads or os.cpu_count()),
                   dict(env or {}, CRH_DUMP_F32=f32, CRH_DUMP_STATS=stats, CRH_NO_IMAGE="1"), timeout,
                   extra_args=("--iterative",) if iterative else ())
        buf = np.fromfile(f32, dtype=np.float32).reshape(height, width, 3)
        with open(stats) as f:
            st = json.load(f)
        st["log_tail"] = log[-400:]
    return buf, st


def flatten_scene(scene_name, out_blob, width, height, samples, bounces, tile=None, timeout=3600, env=None):
    """Run crh-flatten (reference loader + product flattener) to write a scene blob."""
    scene = rewrite_scene(scene_name, width, height, samples, bounces, tile=tile)
    os.makedirs(os.path.dirname(os.path.abspath(out_blob)), exist_ok=True)
    return _run("crh-flatten", scene, None, dict(env or {}, CRH_DUMP_SCENE=os.path.abspath(out_blob)), timeout)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=200)
    ap.add_argument("--samples", type=int, default=4)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--flavour", default="strict")
    ap.add_argument("--blob")
    ap.add_argument("--out")
    a = ap.parse_args()
    if a.blob:
        print(flatten_scene(a.scene, a.blob, a.width, a.height, a.samples, a.bounces)[-600:])
    else:
        buf, st = render_reference(a.scene, a.width, a.height, a.samples, a.bounces, a.flavour)
        st.pop("log_tail", None)
        print(st, float(buf.mean()))
        if a.out:
            buf.tofile(a.out)
