"""Re-exports the synthetic-input generators (they live in the product package so
that bench.py's GPU leg does not have to import anything under oracle/)."""
from scnerf_amd.synthetic import *  # noqa: F401,F403
from scnerf_amd.synthetic import (camera_spec, keypoints, network_params, ray_batch,  # noqa: F401
                                  render_randoms, target_rgb, xavier_nerf_params)
