"""Copy the reference's own DATA files for bgdehaze -- the three input photographs and the one output image it
ships -- into tests/golden/real/ (they are data, not source: inputs and an expected output), and record their
SHA-256.  Run once in the build container:  python tools/make_real_fixtures.py

  /root/reference/modules/bgdehaze/img/{BUL_T1A_0028,BUL_T1A_0209,PIS_T1A_259}.jpg   inputs (1920x1080)
  /root/reference/modules/bgdehaze/result/BUL_T1A_0209.jpg                            main.py's output for the 2nd one
                                                                                      (adaptiveExp_map, w = 15, JPEG)
The tests decode them with Pillow (present in the image, here and on the GPU box); cv2.imread would use libjpeg too,
so the decoded pixels are the same up to the decoder's IDCT / chroma-upsampling choices."""
import hashlib
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/modules/bgdehaze"
OUT = os.path.join(ROOT, "tests", "golden", "real")

FILES = [("../aclahe/python/crowd.png", "in_aclahe_crowd.png"),      # the image modules/aclahe/python/main.py reads (800x600, 8-bit grey)
         ("img/BUL_T1A_0028.jpg", "in_BUL_T1A_0028.jpg"), ("img/BUL_T1A_0209.jpg", "in_BUL_T1A_0209.jpg"),
         ("img/PIS_T1A_259.jpg", "in_PIS_T1A_259.jpg"), ("result/BUL_T1A_0209.jpg", "ref_result_BUL_T1A_0209.jpg")]


def main():
    os.makedirs(OUT, exist_ok=True)
    man = {}
    for src, dst in FILES:
        shutil.copyfile(os.path.join(REF, src), os.path.join(OUT, dst))
        os.chmod(os.path.join(OUT, dst), 0o644)
        man[dst] = {"from": os.path.normpath("modules/bgdehaze/" + src), "sha256": hashlib.sha256(open(os.path.join(OUT, dst), "rb").read()).hexdigest()}
    json.dump(man, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(man, indent=1))


if __name__ == "__main__":
    main()


def crops():
    """The reference also ships two LOSSLESS outputs: result/restoredFiltered.png = main.py on img/BUL_T1A_0028.jpg and
    result/restoredFiltered2.png = main.py on img/PIS_T1A_259.jpg (identified by content: tests/test_real_images.py).
    They are 3.7 + 1.6 MB; the central 960 x 540 window of each is kept (PNG, lossless) -- a quarter of the pixels is
    plenty for per-pixel agreement statistics."""
    from PIL import Image
    y0, y1, x0, x1 = 270, 810, 480, 1440
    for src, dst in (("result/restoredFiltered.png", "ref_result_BUL_T1A_0028_crop.png"), ("result/restoredFiltered2.png", "ref_result_PIS_T1A_259_crop.png")):
        im = Image.open(os.path.join(REF, src)).convert("RGB").crop((x0, y0, x1, y1))
        im.save(os.path.join(OUT, dst), optimize=True)
    json.dump({"window_rows": [y0, y1], "window_cols": [x0, x1], "files": {"ref_result_BUL_T1A_0028_crop.png": "in_BUL_T1A_0028.jpg",
               "ref_result_PIS_T1A_259_crop.png": "in_PIS_T1A_259.jpg"}}, open(os.path.join(OUT, "crops.json"), "w"), indent=1, sort_keys=True)


def b9_indices():
    """Background_light takes `argsort(axis=0)[:1]` of the two difference maps (BGDehaze.py:24), whose
    order among tied minima is whatever numpy's unstable sort yields.  Record, for the one image whose reference
    output exists, which tied pixels THIS container's numpy picks (the difference maps come from our oracle; only
    np.argsort itself is numpy's): injecting the background light of these two pixels reproduces the reference's
    shipped result to JPEG noise (tests/test_real_images.py), the first-index rule of the product does not."""
    import sys
    import numpy as np
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dehaze_oracle as dz
    img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(OUT, "in_BUL_T1A_0209.jpg")).convert("RGB"))[:, :, ::-1])
    normI = dz.normalize_input(img)
    mx = [dz._window_reduce(normI[:, :, c], 15, np.maximum).ravel() for c in range(3)]
    D0, D1 = mx[2] - mx[0], mx[2] - mx[1]
    i0 = int(np.argsort(D0.reshape(-1, 1), axis=0)[:1][0, 0])
    i1 = int(np.argsort(D1.reshape(-1, 1), axis=0)[:1][0, 0])
    flat = normI.reshape(-1, 3)
    d = {"image": "in_BUL_T1A_0209.jpg", "numpy": np.__version__, "argsort_indices": [i0, i1],
         "tie_counts": [int((D0 == D0.min()).sum()), int((D1 == D1.min()).sum())],
         "B_argsort": [float(v) for v in (flat[i0] + flat[i1]) / 2.0]}
    json.dump(d, open(os.path.join(OUT, "b9_argsort.json"), "w"), indent=1, sort_keys=True)
    print(d)


if __name__ == "__main__":
    crops()
    b9_indices()
