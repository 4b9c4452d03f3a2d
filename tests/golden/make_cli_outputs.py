#!/usr/bin/env python3
"""Runs ON THE GPU BOX (gpurun): localises the queries of tests/consumer_scene.py with bin/OpenMVGLocalization_AKAZE
(and with its Python mirror engine.main, which must write the same bytes) and leaves the result files under
gpurun_out/ref_consumers_cli/.  tests/golden/make_ref_fixtures.py (build container, where /root/reference is) then
feeds them to the reference's own consumers and commits files + what the consumers returned.

    gpurun -- python tests/golden/make_cli_outputs.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import consumer_scene as scene
    from sfmlocalization_amd import engine, hulo
    out = os.path.join(ROOT, "gpurun_out", "ref_consumers_cli")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    with tempfile.TemporaryDirectory() as td:
        scene.build(td)
        r = subprocess.run([hulo.LOCALIZE_PROJECT_PATH] + scene.args(), cwd=td, capture_output=True, text=True)
        sys.stdout.write(r.stdout[-2000:])
        sys.stderr.write(r.stderr[-2000:])
        assert r.returncode == 0
        cwd = os.getcwd()
        os.chdir(td)
        try:
            a = scene.args()
            a[3] = "loc_py"
            assert engine.main(a) == 0
        finally:
            os.chdir(cwd)
        for base in scene.QUERY_BASES:
            src = os.path.join(td, "loc", base + ".json")
            with open(src, "rb") as f1, open(os.path.join(td, "loc_py", base + ".json"), "rb") as f2:
                assert f1.read() == f2.read(), f"{base}: the C++ tool and engine.main wrote different bytes"
            shutil.copy(src, os.path.join(out, base + ".json"))
            print(base, os.path.getsize(src), "bytes")


if __name__ == "__main__":
    main()
