"""Image-decode worker PROCESS of the dataset drivers (eva_vos_amd.fq_driver.decode_pool): started as
``python -m eva_vos_amd._decode_worker``, it reads one JSON request per line on stdin, decodes the JPEG frames / palette-PNG label maps it
names straight into shared-memory blocks of the parent (``/dev/shm/<name>``, mapped here with plain mmap: no resource tracker involved) and
answers ``ok`` / ``err: ...`` on stdout.  Imports NumPy and Pillow only - no torch, no HIP: a worker is up in ~0.2 s.

Why processes: Pillow decodes in 64 KB steps driven from Python and ``np.asarray(image)`` goes through ``tobytes()`` in a Python loop, so
decode THREADS take the interpreter lock thousands of times per video; with 16 of them beside the lanes that drive the GPU, every stretch
of Python on a lane (building an InferenceCore, a round's bookkeeping) waited 30-50x longer than it runs (tools/driver_lanes.py, round 6).
Request: {"rgb": name, "lab": name, "shape": [T, H, W], "frames": [[t, jpg_path, png_path], ...]} - either path may be null."""
import json
import mmap
import os
import sys


def main():
    import numpy as np
    from PIL import Image
    out = sys.stdout
    for line in sys.stdin:
        try:
            req = json.loads(line)
            T, H, W = req["shape"]
            maps = []

            def block(name, nbytes):
                fd = os.open(os.path.join("/dev/shm", name), os.O_RDWR)
                try:
                    m = mmap.mmap(fd, nbytes)
                finally:
                    os.close(fd)
                maps.append(m)
                return m

            rgb = np.frombuffer(block(req["rgb"], T * H * W * 3), np.uint8).reshape(T, H, W, 3) if req.get("rgb") else None
            lab = np.frombuffer(block(req["lab"], T * H * W), np.uint8).reshape(T, H, W) if req.get("lab") else None
            for t, jpg, png in req["frames"]:
                if jpg is not None:
                    rgb[t] = np.asarray(Image.open(jpg).convert("RGB"))
                if png is not None:
                    lab[t] = np.array(Image.open(png).convert("P"), dtype=np.uint8)
            del rgb, lab
            for m in maps:
                m.close()
            out.write("ok\n")
        except Exception as ex:                                   # the parent raises with this text
            out.write("err: " + f"{type(ex).__name__}: {ex}".replace("\n", " ") + "\n")
        out.flush()


if __name__ == "__main__":
    main()
