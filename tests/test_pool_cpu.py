"""Size classes of the device-block cache (mcptam_amd/csrc/ba_pool.h, DevCache::class_bytes): every request is rounded up to
(4 + k) * 2^m, k in 0..3 -- at least the request, at most 25 % above it, at least 512 bytes, monotone, and a class's own size maps
to itself (what a handle hands back is what the next one asks for).  Host logic only: runs without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_size_classes(tmp_path):
    exe = str(tmp_path / "pool_classes")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950",
                           os.path.join(ROOT, "tests", "cpp", "pool_classes.hip"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    rows = [tuple(int(v) for v in ln.split()) for ln in out.stdout.strip().splitlines()]
    assert len(rows) >= 15
    last_b, last_c = 0, -1
    for req, b, c in rows:
        assert b >= max(req, 512), (req, b)
        assert b <= max(512, req + req // 4 + 1), (req, b)          # <= 25 % slack
        m = b.bit_length() - 1
        assert b % (1 << (m - 2)) == 0 and c == 4 * m + (b - (1 << m)) // (1 << (m - 2)), (req, b, c)      # (4 + k) * 2^(m-2), class = 4 m + k
        assert b >= last_b and c >= last_c, "monotone"
        last_b, last_c = b, c
    by_req = {r: (b, c) for r, b, c in rows}
    assert by_req[512] == (512, 36) and by_req[513][0] == 640 and by_req[640][0] == 640 and by_req[641][0] == 768
    assert by_req[65536][0] == 65536 and by_req[65537][0] == 81920
