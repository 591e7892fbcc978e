"""Golden bytes for the Gipuma exchange formats: run the reference's own
CasMVSNet/gipuma.py functions (write_gipuma_dmb, read_gipuma_dmb, mvsnet_to_gipuma_cam,
fake_gipuma_normal, probability_filter's thresholding) on small seeded inputs in the build container
and store inputs + produced file bytes.   python tests/golden/make_golden_gipuma.py
Only data is stored."""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from io_fixture import cam_text  # noqa: E402
from make_golden import save  # noqa: E402

REF = "/root/reference/CasMVSNet"


def main():
    for s in ("torchvision", "torchvision.utils", "cv2", "tensorboardX"):
        sys.modules.setdefault(s, types.ModuleType(s))
    for k in [k for k in sys.modules if k in ("utils", "datasets") or k.startswith("datasets.") or k.startswith("models")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    import gipuma as ref
    sys.path.pop(0)
    rng = np.random.default_rng(7)
    d = tempfile.mkdtemp()
    depth = (rng.random((5, 7)) * 500 + 400).astype(np.float32)
    depth[1, 2] = 0.0
    depth[4, 6] = 0.0
    p_depth = os.path.join(d, "disp.dmb")
    ref.write_gipuma_dmb(p_depth, depth)
    p_normal = os.path.join(d, "normals.dmb")
    ref.fake_gipuma_normal(p_depth, p_normal)
    back = ref.read_gipuma_dmb(p_depth)
    normal_back = ref.read_gipuma_dmb(p_normal)
    cam_in = os.path.join(d, "00000002_cam.txt")
    with open(cam_in, "w") as f:
        f.write(cam_text(2))
    cam_out = os.path.join(d, "00000002.jpg.P")
    ref.mvsnet_to_gipuma_cam(cam_in, cam_out)
    save("g15_gipuma", depth=depth, depth_dmb=np.frombuffer(open(p_depth, "rb").read(), dtype=np.uint8),
         normal_dmb=np.frombuffer(open(p_normal, "rb").read(), dtype=np.uint8), depth_back=back,
         normal_back=normal_back, cam_txt=np.frombuffer(open(cam_in, "rb").read(), dtype=np.uint8),
         cam_P=np.frombuffer(open(cam_out, "rb").read(), dtype=np.uint8))
    print(open(cam_out).read())


if __name__ == "__main__":
    main()
