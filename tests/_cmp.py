"""Map comparison helpers: device download (reference record format) vs oracle dump."""
import numpy as np


def diff_maps(dev, orc, fields):
    """dev/orc: {patch_id: (cells[1024] structured, mask[16])}.  Returns a dict of mismatch counts."""
    out = {"patches_only_dev": 0, "patches_only_orc": 0, "mask_words": 0}
    for f in fields:
        out[f] = 0
    for pid in dev.keys() - orc.keys():
        out["patches_only_dev"] += 1
    for pid in orc.keys() - dev.keys():
        out["patches_only_orc"] += 1
    for pid in dev.keys() & orc.keys():
        dc, dm = dev[pid]
        oc, om = orc[pid]
        out["mask_words"] += int((dm != om).sum())
        for f in fields:
            a, b = dc[f], oc[f]
            ne = (a != b)
            if ne.ndim > 1:
                ne = ne.any(axis=1)
            out[f] += int(ne.sum())
    return out


DM_FIELDS = ["sqdist", "valid", "obstacle", "queued"]
OCC_FIELDS = ["occupied", "visited"]


def assert_maps_equal(dev, orc, fields, what=""):
    d = diff_maps(dev, orc, fields)
    assert all(v == 0 for v in d.values()), f"{what}: {d} (dev patches {len(dev)}, oracle patches {len(orc)})"
