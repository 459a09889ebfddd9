"""Hot-path subset of the reference's ``sunflower/utils/image_manipulation.py``:
``shrink_mask`` (:21-36) and ``get_depth_value`` (:39-96), computed on the GPU
(flope_depth_lift) behind the reference's numpy signature."""
import numpy as np
import torch

from flope_amd import engine as _engine


def get_depth_value(bbox, depth, seg_mask, scale=None, near_plane: float = 0.1, far_plane: float = 3.0,
                    vis: bool = False):
    """bbox int [N,4] (xmin,ymin,xmax,ymax), depth float [H,W] metres, seg_mask uint8 [H,W]
    -> (depth_values [N] float64 metres, depth_reliable [N] bool, None).

    valid = (near < depth < far) & (mask > 128), eroded by the 10x10 ellipse; per box the mean
    of the valid depths (0 when empty) and ``count >= 50`` as the reliability flag.
    Unlike the reference the caller's ``depth`` array is not modified in place.
    """
    if vis:
        raise NotImplementedError("depth visualisation (vis=True) is outside the MI355X hot path")
    d = np.ascontiguousarray(depth, dtype=np.float32)
    if scale:
        d = d * np.float32(scale)
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if dev is None:
        raise RuntimeError("flope_amd: no HIP device visible; there is no CPU fallback")
    dv, rel, _ = _engine.depth_lift(torch.from_numpy(d).to(dev),
                                    torch.from_numpy(np.ascontiguousarray(seg_mask, dtype=np.uint8)).to(dev),
                                    torch.from_numpy(np.ascontiguousarray(bbox).astype(np.int32)).to(dev),
                                    (1.0, 1.0, 0.0, 0.0), 1.0, near_plane, far_plane)
    return dv.double().cpu().numpy(), rel.cpu().numpy(), None
