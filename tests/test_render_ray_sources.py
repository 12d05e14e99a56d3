"""render()'s ray-source selection (scnerf_amd/render.py `_select_rays`, a table in place of the reference's five-way
if-chain, /root/reference NeRF/render.py:27-103): every branch picks the generator, the pose and the NDC focal the
reference picks, every precondition the reference asserts raises AssertionError here too, and a mode outside
train / val / test without precomputed rays trips the same "should not appear" assertion.  The ray generators are
replaced by recorders: no kernel runs."""
import numpy as np
import pytest
import torch

from scnerf_amd import render as R


@pytest.fixture
def calls(monkeypatch):
    log = []

    def use_camera(H, W, camera_model, extrinsic=None, **kw):
        log.append(("use_camera", camera_model, extrinsic))
        return torch.zeros(H * W, 3), torch.ones(H * W, 3)

    def no_camera(H, W, focal, extrinsic, **kw):
        log.append(("no_camera", focal, extrinsic))
        return torch.zeros(H * W, 3), torch.ones(H * W, 3)

    monkeypatch.setattr(R, "get_rays_full_image_use_camera", use_camera)
    monkeypatch.setattr(R, "get_rays_full_image_no_camera", no_camera)
    return log


def select(**kw):
    a = dict(H=2, W=3, rays=None, noisy_focal=None, noisy_extrinsic=None, mode=None, camera_model=None, image_idx=None,
             i_map=None, gt_intrinsic=None, gt_extrinsic=None, transform_align=None)
    a.update(kw)
    return R._select_rays(a)


CAM = object()
POSES = [torch.full((3, 4), float(i)) for i in range(6)]


def test_precomputed_rays_win_over_everything(calls):
    ro, rd = torch.rand(5, 3), torch.rand(5, 3)
    o, d, focal = select(rays=(ro, rd), mode="train", noisy_focal=123.0)
    assert o is ro and d is rd and focal == 123.0 and not calls            # pinhole: the noisy focal feeds the NDC warp
    o, d, focal = select(rays=(ro, rd), mode="whatever", camera_model=CAM, noisy_focal=123.0)
    assert focal is None                                                     # camera model: its own focal lengths


def test_trained_camera_train_view(calls):
    i_map = np.array([4, 7, 9])
    _, _, focal = select(camera_model=CAM, mode="train", i_map=i_map, image_idx=7, noisy_extrinsic=POSES)
    assert calls == [("use_camera", CAM, POSES[1])] and focal is None        # slot of image 7 in i_map
    for bad in (dict(i_map=None), dict(image_idx=5), dict(gt_intrinsic=torch.eye(3)), dict(gt_extrinsic=POSES)):
        kw = dict(camera_model=CAM, mode="train", i_map=i_map, image_idx=7, noisy_extrinsic=POSES)
        kw.update(bad)
        with pytest.raises(AssertionError):
            select(**kw)


@pytest.mark.parametrize("mode", ["val", "test"])
def test_trained_camera_held_out_view(calls, mode):
    align = torch.eye(4)
    select(camera_model=CAM, mode=mode, transform_align=align)
    assert calls == [("use_camera", CAM, align)]
    for bad in (dict(noisy_focal=1.0), dict(noisy_extrinsic=POSES)):
        with pytest.raises(AssertionError):
            select(camera_model=CAM, mode=mode, transform_align=align, **bad)


def test_noisy_pinhole_train_view(calls):
    _, _, focal = select(mode="train", noisy_focal=400.0, noisy_extrinsic=POSES, image_idx=3)
    assert calls == [("no_camera", 400.0, POSES[3])] and focal == 400.0
    for bad in (dict(noisy_focal=None), dict(noisy_extrinsic=None)):
        kw = dict(mode="train", noisy_focal=400.0, noisy_extrinsic=POSES, image_idx=3)
        kw.update(bad)
        with pytest.raises(AssertionError):
            select(**kw)


@pytest.mark.parametrize("mode", ["val", "test"])
def test_ground_truth_pinhole(calls, mode):
    K = torch.tensor([[555.0, 0, 1], [0, 555.0, 1], [0, 0, 1]])
    _, _, focal = select(mode=mode, gt_intrinsic=K, gt_extrinsic=POSES, image_idx=2)
    assert calls == [("no_camera", 555.0, POSES[2])] and focal == 555.0
    for bad in (dict(gt_extrinsic=None), dict(noisy_focal=1.0), dict(noisy_extrinsic=POSES)):
        kw = dict(mode=mode, gt_intrinsic=K, gt_extrinsic=POSES, image_idx=2)
        kw.update(bad)
        with pytest.raises(AssertionError):
            select(**kw)


def test_unknown_mode_without_rays_is_the_failure_branch(calls):
    with pytest.raises(AssertionError, match="should not appear"):
        select(mode="render_only")
    with pytest.raises(AssertionError, match="should not appear"):
        select(mode="render_only", camera_model=CAM)
    with pytest.raises(AssertionError):                                       # render() itself: mode is mandatory
        R.render(2, 3, 8, rays=(torch.rand(1, 3), torch.rand(1, 3)))
