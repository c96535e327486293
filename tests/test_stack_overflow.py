"""Deep trees are traversed with 22 stack entries per lane in LDS and the rest in private memory (LdsStack's
overflow entries, csrc/ptk_kernels.hip).  A lane needs more than twenty pending entries next to never, so the
ordinary build hardly ever takes that path; tests/variants/libptk_shortstack.so is the same library built with
SIX LDS entries, which sends every tree of more than four levels through the overflow entries all the time.
Where a stack entry lives must not change a bit: the short-stack build has to give the frames and the hits of
the ordinary build."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.backends import GpuBackend, scene
from tests import raygen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHORT = os.path.join(ROOT, "tests", "variants", "libptk_shortstack.so")
CASES = [("mesh", 16, 3), ("glass", 8, 8), ("sky", 8, 4)]


def _digests(ctx):
    out = {}
    for name, spp, bounces in CASES:
        b = GpuBackend(ctx, scene(name))
        img, cnt = b.render(spp, bounces, seed=9, collect_stats=True)
        rays = np.concatenate([raygen.interior_rays(20000, 7), raygen.camera_like_rays(10000, 8)]).astype(np.float32).astype(np.float64)
        hits, _ = b.intersect(rays)
        out[name] = dict(frame=hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest(),
                         ids=hashlib.sha256(np.ascontiguousarray(hits["id"]).tobytes()).hexdigest(),
                         t=hashlib.sha256(np.ascontiguousarray(hits["t"]).tobytes()).hexdigest(),
                         depth=b.info()["max_depth"], counters=cnt)
    return out


def test_short_stack_build_gives_the_same_frames_and_hits(gpu_ctx):
    if not os.path.exists(SHORT):
        pytest.fail(f"{SHORT} is missing: build it with `make -C pathtracer_amd/csrc shortstack`")
    mine = _digests(gpu_ctx)
    assert all(v["depth"] > 4 for v in mine.values())  # deeper than the short build's LDS part can hold
    env = dict(os.environ, PTK_LIB=SHORT, PYTHONPATH=ROOT)
    code = ("import json; from pathtracer_amd import native; from tests.test_stack_overflow import _digests; "
            "ctx = native.Context(0); print('DIGESTS ' + json.dumps(_digests(ctx))); ctx.close()")
    run = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    line = [l for l in run.stdout.splitlines() if l.startswith("DIGESTS ")][-1]
    theirs = json.loads(line[len("DIGESTS "):])
    assert theirs == json.loads(json.dumps(mine))
