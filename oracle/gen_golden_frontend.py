"""Writes tests/golden/frontend_pil.npz: Pillow's own BICUBIC output (the library the reference calls at
src/data.py:93-96) for a few crops, plus the inputs, and the reference's own `tif_image` / `padded_crop`
(src/util/geo_util.py:449-470, 316-341; imported with the absent third-party modules stubbed exactly as
oracle/gen_golden.py does -- none of them is on these two functions' arithmetic path).
Run in the build container only (needs PIL and /root/reference)."""
import sys
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
from PIL import Image

root = Path(__file__).resolve().parents[1]
rng = np.random.default_rng(2024)
mosaic = rng.integers(0, 256, size=(300, 260, 3), dtype=np.uint8)
yy, xx = np.mgrid[0:300, 0:260]
mosaic[..., 1] = ((yy * 3 + xx * 5) % 256).astype(np.uint8)           # a ramp band: exercises the clip8 edges less randomly
mosaic[100:140, 60:120] = 255                                          # saturated block: overshoot must clip at 255
mosaic[180:200, 10:50] = 0
boxes = np.array([[37, 91, 149, 203], [200, 250, 312, 362]], dtype=np.int32)  # interior; clipped at both mosaic edges
boxes256 = np.array([[60, 100, 316, 356]], dtype=np.int32)


def pil(box, crop, S):
    x0, y0 = box[0], box[1]
    win = np.zeros((crop, crop, 3), dtype=np.uint8)
    ys, xs = min(mosaic.shape[0], y0 + crop) - y0, min(mosaic.shape[1], x0 + crop) - x0
    win[:ys, :xs] = mosaic[y0:y0 + ys, x0:x0 + xs]
    return np.array(Image.fromarray(win).resize((S, S), resample=Image.Resampling.BICUBIC))


out112 = np.stack([pil(b, 112, 448) for b in boxes])
out256 = np.stack([pil(b, 256, 448) for b in boxes256])
down = np.stack([pil(b, 256, 96) for b in boxes256])                    # down-scaling: wider kernels (support 2 * scale)
# ---- the reference's 4-band -> RGB collapse and padded crop
for name in ["geopandas", "rasterio", "rasterio.features", "rasterio.merge", "rasterio.warp", "rasterio.transform",
             "rasterio.io", "rasterio.enums", "rasterio.crs", "rasterio.windows", "rasterio.mask", "shapely", "shapely.geometry",
             "shapely.ops", "skimage", "skimage.morphology", "skimage.measure", "skimage.graph", "affine", "cv2",
             "matplotlib", "matplotlib.pyplot", "matplotlib.patches", "matplotlib.colors", "matplotlib.axes", "pyproj"]:
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = MagicMock()
sys.path.insert(0, "/root/reference")
from src.util import geo_util as ref_geo  # noqa: E402

bands = rng.uniform(200, 3200, size=(4, 48, 40)).astype(np.float32)  # SR-like Dove tile (SURVEY section 8 d, config 1)
bands[:, 5:9, 7:30] += 4000.0                                          # bright strip: exercises the [min, min+3000] clip
nodata = np.zeros((48, 40), dtype=bool)
nodata[40:, :6] = True
tif_rgb = ref_geo.tif_image(bands.copy(), nodata)
# 8-band (SuperDove) branch: src/util/multichannel_img.py:7-29 through the same entry point
bands8 = rng.uniform(150, 4200, size=(8, 40, 56)).astype(np.float32)
nodata8 = np.zeros((40, 56), dtype=bool)
nodata8[:5, 50:] = True
tif8_rgb = ref_geo.tif_image(bands8.copy(), nodata8)
pc_src = np.arange(7 * 9 * 3, dtype=np.uint8).reshape(7, 9, 3)
pc_boxes = np.array([[-2, -1, 3, 4], [6, 4, 11, 9], [2, 1, 7, 6]], dtype=np.int32)
pc_out = np.stack([ref_geo.padded_crop(pc_src, int(b[0]), int(b[1]), int(b[2]), int(b[3]), 5, 0) for b in pc_boxes])

np.savez_compressed(root / "tests" / "golden" / "frontend_pil.npz", tif_bands=bands, tif_nodata=nodata, tif_rgb=tif_rgb,
                    tif8_bands=bands8, tif8_nodata=nodata8, tif8_rgb=tif8_rgb,
                    pc_src=pc_src, pc_boxes=pc_boxes, pc_out=pc_out, mosaic=mosaic, boxes112=boxes, out112=out112,
                    boxes256=boxes256, out256=out256, down256_96=down, pil_version=np.array(Image.__version__))
print("wrote frontend_pil.npz", out112.shape, out256.shape, down.shape, Image.__version__)
