"""The C++ host mirror (include/ktlvo_host.hpp): a node written like the reference's own callback,
call by call, over the mirror's functions (tests/cpp/frontend_node.cpp) must publish the very bytes
the library's one-call frame path and this harness's ctypes path publish, keep the features the CPU
oracle keeps, and throw where OpenCV would assert."""
import os
import subprocess

import numpy as np
import pytest

from slam_amd import synth

pytestmark = pytest.mark.gpu

NODE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "frontend_node")


def test_reference_shaped_node_matches_frame_path_and_oracle(tmp_path, make_ctx, klib):
    assert os.path.exists(NODE), "tests/cpp/frontend_node is not built (python -c 'import __graft_entry__ as g; g.build()')"
    W, H, NF = 1242, 375, 8
    scene = synth.CorridorScene(W, H, seed=0x5EED0001)
    pairs = [scene.stereo_pair(f) for f in range(NF)]
    raw = tmp_path / "frames.bin"
    with open(raw, "wb") as fh:
        for L, R in pairs:
            fh.write(np.ascontiguousarray(L).tobytes())
            fh.write(np.ascontiguousarray(R).tobytes())
    outp = tmp_path / "out.bin"
    r = subprocess.run([NODE, str(raw), str(W), str(H), str(NF), str(outp)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout, r.stdout
    print(r.stdout)

    # the node's output against the ctypes path of this process
    blob = np.fromfile(outp, np.uint8)
    ctx = make_ctx()
    off, total = 0, 0
    for f, (L, R) in enumerate(pairs):
        n, key = np.frombuffer(blob, np.int32, 2, off)
        off += 8
        last = np.frombuffer(blob, klib.DEPTH_POINT, n, off)
        off += 24 * int(n)
        cur = np.frombuffer(blob, klib.IMAGE_POINT, n, off)
        off += 12 * int(n)
        Tg = np.frombuffer(blob, np.float32, 16, off).reshape(4, 4)
        off += 64
        gl, gc, gkf = ctx.track_stereo_frame(L, R)
        assert bool(key) == bool(gkf) and len(gl) == n, f
        assert gl.tobytes() == last.tobytes() and gc.tobytes() == cur.tobytes(), f
        _, _, gTg = ctx.estimate_pose(gl, gc)
        assert np.array_equal(gTg, Tg), f
        total += int(n)
    assert off == blob.size and total > 1000
