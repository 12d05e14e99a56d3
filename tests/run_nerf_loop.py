"""TEST INFRASTRUCTURE: a training driver written against the *mirrored API* in the order and with the
keyword sets of the reference's training script (/root/reference NeRF/run_nerf.py:76-1040), for the GPU
box, where the reference tree does not exist.

It is not a copy of that script: data handling, logging, evaluation metrics and the matchers are left
out, and what remains is organised as small steps of one `Loop` object.  What ties it to the reference is
mechanical: tests/test_dropin_run_nerf.py extracts every call the reference's `train()` makes into the
mirrored modules (callee, number of positional arguments, keyword names) and requires this file to make
the same calls (`tests/golden/run_nerf_calls.json`, regenerated and compared with the live reference
where it is available).  The same test file runs the UNMODIFIED run_nerf.py itself -- on the CPU SIMT
interpreter in the build container, on the GPU wherever a reference checkout and a GPU coexist."""
import os
import types

import numpy as np
import torch

from scnerf_amd.create_nerf import create_nerf
from scnerf_amd.get_rays import get_rays_kps_no_camera, get_rays_kps_use_camera, get_rays_np
from scnerf_amd.camera_model import *                      # noqa: F401,F403  (the script's star import: wandb, np, torch ...)
from scnerf_amd.prd_evaluation import projected_ray_distance_evaluation
from scnerf_amd.ray_dist_loss import preprocess_match, proj_ray_dist_loss_single
from scnerf_amd.render import render, render_path
from scnerf_amd.run_nerf_helpers import fix_seeds, img2mse, mse2psnr


class Loop:
    """One training run: `Loop(args, data, device).run()`; `history` collects what the script would log."""

    def __init__(self, args, data, device, matcher=None):
        self.args, self.device, self.matcher = args, device, matcher
        images, poses, bds, render_poses, i_test, (gt_intrinsic, gt_extrinsic) = data
        self.i_test = self.i_val = np.asarray(i_test)
        self.i_train = np.array([i for i in range(images.shape[0]) if i not in self.i_test])
        hwf = poses[self.i_train[0], :3, -1]
        self.H, self.W, self.noisy_focal = int(hwf[0]), int(hwf[1]), hwf[2]
        ext = np.zeros((len(poses), 4, 4), np.float32)
        ext[:, :3, :] = poses[:, :3, :4]
        ext[:, 3, 3] = 1
        self.noisy_extrinsic_np = ext
        self.gt_intrinsic, self.gt_extrinsic = gt_intrinsic, gt_extrinsic
        self.render_poses = render_poses
        self.near, self.far = (float(bds.min()) * .9, float(bds.max())) if args.no_ndc else (0., 1.)
        self.images_np = images
        self.history = []
        self.image_pair_cache = {}

    # ---- set-up (run_nerf.py:76-345) -----------------------------------------------------------------
    def setup(self):
        args, H, W = self.args, self.H, self.W
        fix_seeds(args.seed)
        os.makedirs(os.path.join(args.basedir, args.expname), exist_ok=True)
        noisy_train_poses = self.noisy_extrinsic_np[self.i_train]
        (self.render_kwargs_train, self.render_kwargs_test, start, grad_vars, self.optimizer,
         self.camera_model) = create_nerf(args, self.noisy_focal, noisy_train_poses, H, W, mode="train",
                                          device=self.device)
        self.global_step = start
        self.start = start + 1
        bounds = {'near': self.near, 'far': self.far}
        self.render_kwargs_train.update(bounds)
        self.render_kwargs_test.update(bounds)
        self.use_batching = not args.no_batching
        if self.use_batching:
            if self.camera_model is None:
                rays = np.stack([get_rays_np(H, W, self.noisy_focal, p) for p in self.noisy_extrinsic_np[:, :3, :4]], 0)
                rays_rgb = np.concatenate([rays, self.images_np[:, None]], 1).transpose(0, 2, 3, 1, 4)
                self.rays_rgb = np.stack([rays_rgb[i] for i in self.i_train], 0).reshape(-1, 3, 3).astype(np.float32)
            self.shuffled_ray_idx = np.arange(len(self.i_train) * H * W)
            np.random.shuffle(self.shuffled_ray_idx)
            self.shuffled_image_idx = self.shuffled_ray_idx // (H * W)
            if self.camera_model is None:
                self.rays_rgb = torch.tensor(self.rays_rgb[self.shuffled_ray_idx]).to(self.device)
            self.i_batch = 0
        self.images = torch.tensor(self.images_np).to(self.device)
        self.noisy_extrinsic = torch.tensor(self.noisy_extrinsic_np).to(self.device)
        if args.ray_loss_type != "none":
            # the reference asks image_pair_candidates (matcher side); here every other train view is a candidate
            self.image_pairs = {int(i): [int(j) for j in self.i_train if j != i] for i in self.i_train}
        return self

    # ---- curriculum (:346-368) -----------------------------------------------------------------------
    def curriculum(self, i):
        cam, args = self.camera_model, self.args
        if cam is None:
            return
        if i == self.start and i < args.add_ie:
            cam.intrinsics_noise.requires_grad_(False)
            cam.extrinsics_noise.requires_grad_(False)
        if i == self.start and i < args.add_od:
            cam.ray_o_noise.requires_grad_(False)
            cam.ray_d_noise.requires_grad_(False)
        if i == args.add_ie:
            cam.intrinsics_noise.requires_grad_(True)
            cam.extrinsics_noise.requires_grad_(True)
        if i == args.add_od:
            cam.ray_o_noise.requires_grad_(True)
            cam.ray_d_noise.requires_grad_(True)

    # ---- ray batch (:374-478) ------------------------------------------------------------------------
    def ray_batch(self, i):
        args, H, W, N_rand, cam = self.args, self.H, self.W, self.args.N_rand, self.camera_model
        self.img_i = None
        if self.use_batching and cam is None:
            batch = torch.transpose(self.rays_rgb[self.i_batch:self.i_batch + N_rand], 0, 1)
            self.i_batch += N_rand
            if self.i_batch >= self.rays_rgb.shape[0]:
                self.rays_rgb = self.rays_rgb[torch.randperm(self.rays_rgb.shape[0])]
                self.i_batch = 0
            return batch[:2], batch[2]
        if self.use_batching:
            sel = self.shuffled_ray_idx[self.i_batch:self.i_batch + N_rand]
            image_idx = self.shuffled_image_idx[self.i_batch:self.i_batch + N_rand]
            h_list, w_list = sel % (H * W) // W, sel % (H * W) % W
            kps_list = torch.from_numpy(np.stack([w_list, h_list], -1)).to(self.device)
            rays_o, rays_d = get_rays_kps_use_camera(
                H=H, W=W, camera_model=cam, idx_in_camera_param=torch.from_numpy(image_idx).to(self.device),
                kps_list=kps_list)
            index_train = self.i_train[image_idx]
            target_s = self.images[index_train, h_list, w_list]
            self.img_i = np.random.choice(index_train)
            self.i_batch += N_rand
            if self.i_batch >= len(self.shuffled_ray_idx):
                np.random.shuffle(self.shuffled_ray_idx)
                self.shuffled_image_idx = self.shuffled_ray_idx // (H * W)
                self.i_batch = 0
            return torch.stack([rays_o, rays_d]), target_s
        self.img_i = np.random.choice(self.i_train)
        slot = np.where(self.i_train == self.img_i)[0][0]
        target = self.images[self.img_i]
        gx, gy = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
        coords = torch.stack([gx, gy], -1).reshape(-1, 2)
        select_coords = coords[np.random.choice(coords.shape[0], size=[N_rand], replace=False)].long()
        if cam is None:
            rays_o, rays_d = get_rays_kps_no_camera(H=H, W=W, focal=self.noisy_focal,
                                                    extrinsic=self.noisy_extrinsic[self.img_i, :3, :4],
                                                    kps_list=select_coords)
        else:
            rays_o, rays_d = get_rays_kps_use_camera(H=H, W=W, camera_model=cam, idx_in_camera_param=slot,
                                                     kps_list=select_coords)
        return torch.stack([rays_o, rays_d], 0), target[select_coords[:, 1], select_coords[:, 0]]

    # ---- one optimisation step (:480-625) ------------------------------------------------------------
    def step(self, i):
        args, H, W, cam = self.args, self.H, self.W, self.camera_model
        scalars = {}
        batch_rays, target_s = self.ray_batch(i)
        if cam is None:
            rgb, disp, acc, extras = render(H=H, W=W, chunk=args.chunk, noisy_focal=self.noisy_focal,
                                            rays=batch_rays, verbose=i < 10, retraw=True, mode="train",
                                            **self.render_kwargs_train)
        else:
            rgb, disp, acc, extras = render(H=H, W=W, chunk=args.chunk, rays=batch_rays, verbose=i < 10,
                                            retraw=True, camera_model=cam, mode="train",
                                            **self.render_kwargs_train)
        self.optimizer.zero_grad()
        train_loss_1 = img2mse(rgb, target_s)
        train_psnr_1 = mse2psnr(train_loss_1)
        train_loss = train_loss_1
        if 'rgb0' in extras:
            train_loss_0 = img2mse(extras['rgb0'], target_s)
            train_loss = train_loss + train_loss_0
            scalars["train/level_0_psnr"] = mse2psnr(train_loss_0).item()
        prd_due = args.ray_loss_type != "none" and self.global_step >= args.add_prd and (
            self.global_step % args.i_ray_dist_loss == 1 or args.i_ray_dist_loss == 1)
        if prd_due and cam is not None and self.img_i in self.image_pairs:
            ray_dist_loss_ret, n_match = self.prd_term()
            if ray_dist_loss_ret is not None:
                scalars["train/ray_dist_loss"] = ray_dist_loss_ret.item()
                scalars["train/n_match"] = n_match
                train_loss = train_loss + args.ray_dist_loss_weight * ray_dist_loss_ret
        train_loss.backward()
        self.optimizer.step()
        if cam is not None and self.global_step % 2000 == 1:
            scalar_dict, image_dict = cam.log_noises(self.gt_intrinsic, self.gt_extrinsic[self.i_train])
            scalars.update(scalar_dict)
            scalars.update({k: wandb.Image(v) for k, v in image_dict.items()})
        new_lrate = args.lrate * (0.1 ** (self.global_step / (args.lrate_decay * 1000)))
        for param_group in self.optimizer.param_groups:
            param_group['lr'] = new_lrate
        if i % args.i_weights == 0:
            self.save(i)
        scalars.update({"train/loss": train_loss.item(), "train/level_1_psnr": train_psnr_1.item(), "lr": new_lrate})
        wandb.log(scalars)
        self.history.append(scalars)
        self.global_step += 1

    def prd_term(self):
        """The projected-ray-distance term of one image pair (:508-598); matches come from `self.matcher`."""
        args, H, W, cam = self.args, self.H, self.W, self.camera_model
        img_i = self.img_i
        img_j = np.random.choice(self.image_pairs[int(img_i)])
        slot_i, slot_j = np.where(self.i_train == img_i)[0][0], np.where(self.i_train == img_j)[0][0]
        pair_key = (slot_i, slot_j)
        if pair_key in self.image_pair_cache:
            result = self.image_pair_cache[pair_key]
        else:
            with torch.no_grad():
                result = preprocess_match(self.matcher(self.images[img_i], self.images[img_j]))
            if result[0] is not None and result[1] is not None:
                self.image_pair_cache[pair_key] = result
        if result[0] is None or result[1] is None:
            return None, 0
        kps0_list, kps1_list = result
        rays_i = get_rays_kps_use_camera(H=H, W=W, camera_model=cam, idx_in_camera_param=slot_i, kps_list=kps0_list)
        rays_j = get_rays_kps_use_camera(H=H, W=W, camera_model=cam, idx_in_camera_param=slot_j, kps_list=kps1_list)
        return proj_ray_dist_loss_single(kps0_list=kps0_list, kps1_list=kps1_list, img_idx0=img_i, img_idx1=img_j,
                                         rays0=rays_i, rays1=rays_j, mode="train", device=self.device, H=H, W=W,
                                         args=args, camera_model=cam, method="NeRF", i_map=self.i_train)

    # ---- checkpoint (:627-643) -----------------------------------------------------------------------
    def save(self, i):
        path = os.path.join(self.args.basedir, self.args.expname, '{:06d}.tar'.format(i))
        save_dict = {
            'global_step': self.global_step,
            'network_fn_state_dict': self.render_kwargs_train['network_fn'].state_dict(),
            'network_fine_state_dict': self.render_kwargs_train['network_fine'].state_dict(),
            'optimizer_state_dict': self.optimizer.state_dict(),
        }
        if self.args.camera_model != "none":
            save_dict["camera_model"] = self.camera_model.state_dict()
        torch.save(save_dict, path)
        self.last_checkpoint = path

    # ---- evaluation renders (:645-660, :744-790, :826-856, :958-990) ---------------------------------
    def validation_render(self):
        H, W, cam, args = self.H, self.W, self.camera_model, self.args
        img_i = np.random.choice(self.i_val)
        with torch.no_grad():
            if cam is None:
                rgb, disp, acc, extras = render(H=H, W=W, chunk=args.chunk, gt_intrinsic=self.gt_intrinsic,
                                                gt_extrinsic=self.gt_extrinsic, mode="val", image_idx=img_i,
                                                **self.render_kwargs_test)
            else:
                aligned = self.gt_extrinsic[self.i_val]             # the script aligns with ATE first (thirdparty)
                rgb, disp, acc, extras = render(H=H, W=W, chunk=args.chunk, gt_intrinsic=self.gt_intrinsic,
                                                gt_extrinsic=self.gt_extrinsic, mode="val", i_map=self.i_val,
                                                image_idx=img_i, camera_model=cam,
                                                transform_align=aligned[np.where(self.i_val == img_i)[0][0]],
                                                **self.render_kwargs_test)
        rgb = rgb.reshape(H, W, 3)
        return rgb, disp.reshape(H, W), mse2psnr(img2mse(rgb, self.images[img_i])).item()

    def test_render(self, savedir=None):
        cam, args = self.camera_model, self.args
        _hwf = (self.H, self.W, None)
        with torch.no_grad():
            if cam is None:
                return render_path(self.gt_extrinsic[self.i_test], _hwf, args.chunk, self.render_kwargs_test,
                                   gt_imgs=self.images[self.i_test], savedir=savedir, mode="test", args=args,
                                   gt_intrinsic=self.gt_intrinsic, gt_extrinsic=self.gt_extrinsic, i_map=self.i_test)
            aligned = self.gt_extrinsic[self.i_test]
            return render_path(aligned, _hwf, args.chunk, self.render_kwargs_test, gt_imgs=self.images[self.i_test],
                               savedir=savedir, mode="test", camera_model=cam, args=args,
                               gt_intrinsic=self.gt_intrinsic, gt_extrinsic=self.gt_extrinsic, i_map=self.i_test,
                               transform_align=aligned)

    def train_view_render(self, savedir=None):
        """End-of-training rendering of train view 0 (:958-990).  The script sets `i_train = [0]` but still hands
        over the poses of ALL train views, so its render_path indexes i_map[1] and stops with an IndexError
        after the first image whenever there is more than one train view; here the poses are cut to the views
        in i_train, which is what that first image is."""
        cam, args = self.camera_model, self.args
        i_train = [0]
        with torch.no_grad():
            if cam is None:
                return render_path(render_poses=self.noisy_extrinsic[i_train], noisy_extrinsic=self.noisy_extrinsic[i_train],
                                   hwf=[self.H, self.W, self.noisy_focal], chunk=args.chunk,
                                   render_kwargs=self.render_kwargs_train, mode="train", gt_imgs=self.images[i_train],
                                   savedir=savedir, args=args)
            return render_path(render_poses=cam.get_extrinsic()[i_train], noisy_extrinsic=cam.get_extrinsic(),
                               hwf=(self.H, self.W, None), chunk=args.chunk, render_kwargs=self.render_kwargs_train,
                               mode="train", gt_imgs=self.images[i_train], savedir=savedir, camera_model=cam,
                               args=args, i_map=i_train)

    # ---- PRD evaluation (:693-731 test, :879-914 val, :912-951 after training) -----------------------------
    def evaluate_prd(self, mode):
        """mode: "test" / "val" (held-out views through the ground-truth poses) or "train" (the cameras under
        training).  `self.matcher` is handed over as the matcher object; the match FUNCTIONS come from
        scnerf_amd.prd_evaluation._matchers (the reference's reprojection module, or the test's stand-in)."""
        cam, args, H, W = self.camera_model, self.args, self.H, self.W
        if cam is None:
            noisy_K = torch.tensor([[self.noisy_focal, 0, W / 2, 0], [0, self.noisy_focal, H / 2, 0], [0, 0, 1, 0],
                                    [0, 0, 0, 1]], device=self.device)
            intrinsic, extrinsic = (noisy_K, self.noisy_extrinsic) if mode == "train" else (self.gt_intrinsic, self.gt_extrinsic)
            return projected_ray_distance_evaluation(images=self.images, index_list=self.i_train if mode == "train" else self.i_test,
                                                     args=args, ray_fun=get_rays_kps_no_camera, ray_fun_gt=get_rays_kps_no_camera,
                                                     H=H, W=W, mode=mode, matcher=self.matcher, gt_intrinsic=self.gt_intrinsic,
                                                     gt_extrinsic=self.gt_extrinsic, method="NeRF", device=self.device,
                                                     intrinsic=intrinsic, extrinsic=extrinsic)
        if mode == "test":
            return projected_ray_distance_evaluation(images=self.images, index_list=self.i_test, args=args,
                                                     ray_fun=get_rays_kps_use_camera, ray_fun_gt=get_rays_kps_no_camera, H=H, W=W,
                                                     mode="test", matcher=self.matcher, gt_intrinsic=self.gt_intrinsic,
                                                     gt_extrinsic=self.gt_extrinsic, method="NeRF", device=self.device,
                                                     camera_model=cam, intrinsic=self.gt_intrinsic, extrinsic=self.gt_extrinsic)
        if mode == "val":
            return projected_ray_distance_evaluation(images=self.images, index_list=self.i_val, args=args,
                                                     ray_fun=get_rays_kps_use_camera, ray_fun_gt=get_rays_kps_no_camera, H=H, W=W,
                                                     mode="val", matcher=self.matcher, gt_intrinsic=self.gt_intrinsic,
                                                     gt_extrinsic=self.gt_extrinsic, method="NeRF", device=self.device,
                                                     camera_model=cam)
        return projected_ray_distance_evaluation(images=self.images, index_list=self.i_train, args=args,
                                                 ray_fun=get_rays_kps_use_camera, ray_fun_gt=get_rays_kps_no_camera, H=H, W=W,
                                                 mode="train", matcher=self.matcher, gt_intrinsic=self.gt_intrinsic,
                                                 gt_extrinsic=self.gt_extrinsic, method="NeRF", device=self.device,
                                                 camera_model=cam, i_map=self.i_train)

    def render_only(self, savedir=None):
        """`--render_only` (:224-262): the spiral path through the test-time kwargs (only the rotation block of
        each pose is expanded to 4x4 there)."""
        cam, args = self.camera_model, self.args
        poses = self.render_poses.to(self.device)
        render_poses_expand = torch.zeros((len(poses), 4, 4), device=self.device)
        render_poses_expand[:, :3, :3] = poses[:, :3, :3]
        render_poses_expand[:, 3, 3] = 1.0
        _hwf = (self.H, self.W, None)
        with torch.no_grad():
            if cam is None:
                return render_path(render_poses_expand, _hwf, args.chunk, self.render_kwargs_test, savedir=savedir,
                                   mode="test", args=args, gt_intrinsic=self.gt_intrinsic,
                                   gt_extrinsic=render_poses_expand)
            return render_path(render_poses_expand, _hwf, args.chunk, self.render_kwargs_test, savedir=savedir,
                               mode="test", camera_model=cam, args=args, transform_align=render_poses_expand)

    def run(self, n_iters=None):
        self.setup()
        n_iters = self.args.N_iters if n_iters is None else n_iters
        for i in range(self.start, n_iters):
            self.curriculum(i)
            self.step(i)
        return self


def namespace(d):
    return types.SimpleNamespace(**d)
