"""SURVEY.md 8(d)'s end-to-end metric from REAL msdfgen::Shape objects: dumps the bench workload (8 192 distinct DejaVu glyphs, 64x64 frames), has tests/shim/shim_check
rebuild the Shape objects and time msdfgen_hip::generateMSDFBatch() (float tiles, 8-bit atlas).   python tools/e2e_stream.py [reps] [extra env K=V ...]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import load_dejavu  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    for kv in sys.argv[2:]:
        k, v = kv.split("=", 1)
        env[k] = v
    batch, xfs, _ = load_dejavu()
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        path = f.name
    try:
        batch.dump(path, xfs)
        r = subprocess.run([os.path.join(ROOT, "tests", "shim", "shim_check"), "e2e", path, "64", "64", str(reps)], capture_output=True, text=True, env=env, timeout=300)
        sys.stderr.write(r.stderr)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "{}"
        d = json.loads(line)
        d["env"] = {k: v for k, v in (kv.split("=", 1) for kv in sys.argv[2:])}
        print(json.dumps(d), flush=True)
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
