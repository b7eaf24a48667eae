"""TEST INFRASTRUCTURE: bench.py's launcher path on a CPU container.

Installs the host-emulation build of the library (pyshim) and calls bench.main(device="cpu") -- bench.py itself has no way to
load anything but the HIP library.  `python tests/hostemu/bench_emu.py --gpus 2 --backend gloo ...` self-launches its two
ranks exactly as bench.py does on a GPU box (bench.self_launch re-launches the script that was invoked, i.e. this wrapper);
every line produced this way carries an INVALID marker."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

if __name__ == "__main__":
    import bench
    launching = "WORLD_SIZE" not in os.environ and any(a == "--gpus" and sys.argv[i + 1] != "1" for i, a in enumerate(sys.argv[:-1]))
    if not launching:                  # a rank (or a 1-process run): swap the library in, then run the benchmark body
        import pyshim
        pyshim.install()
    bench.main(device="cpu")
