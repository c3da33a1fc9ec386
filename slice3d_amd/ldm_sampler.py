"""DDIM sampling loop of the gen_slices latent-diffusion model (reference: ldm/models/diffusion/ddim.py:10-203, driven by
LatentDiffusion.log_images_when_testing, ddpm.py:449,483: 200 DDIM steps, eta = 1) around slice3d_amd.ldm_unet.UNetModel.

Conditioning as configs/latent-diffusion/objaverse-ldm-kl-8.yaml sets it up: conditioning_key 'concat' — the noisy latent
and the condition latent are concatenated on the channel axis (DiffusionWrapper.forward, ddpm.py:1464-1466) — plus the
feature-map injection c_fmaps of this repository's UNetModel (openaimodel.py:731-746).

The loop runs on fixed shapes, so the ~480 launches of a denoising step are captured ONCE into a HIP graph and replayed
per step (the UNet reads a static input buffer and a static timestep tensor; the x_{t-1} update is four elementwise torch
ops per step on a 16 k-element latent).  Noise comes from a torch.Generator, or from a caller-supplied list (parity tests
feed the reference's draws).
"""
import numpy as np
import torch


def make_beta_schedule(n_timestep=1000, linear_start=0.0015, linear_end=0.0155):
    """'linear' schedule of ldm/modules/diffusionmodules/util.py:21-25 with the yaml's linear_start / linear_end (float64)."""
    return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2


def make_ddim_timesteps(num_ddim_timesteps, num_ddpm_timesteps=1000):
    """util.py:46-60, 'uniform': every c-th step, + 1."""
    c = num_ddpm_timesteps // num_ddim_timesteps
    return np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta):
    """util.py:63-74: sigma_t, alpha_t, alpha_{t-1} of the selected steps."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


class DDIMSampler:
    def __init__(self, unet, timesteps=1000, linear_start=0.0015, linear_end=0.0155, use_graph=True):
        self.unet = unet
        self.num_timesteps = timesteps
        betas = make_beta_schedule(timesteps, linear_start, linear_end)
        # ddpm.py:125-137 registers the cumulative products as float32 buffers; the sampler's numpy maths starts from those
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0).astype(np.float32)
        self.use_graph = use_graph
        self._graph = None
        self._key = None

    def make_schedule(self, ddim_num_steps, ddim_eta=0.0):
        self.ddim_timesteps = make_ddim_timesteps(ddim_num_steps, self.num_timesteps)
        sig, a, ap = make_ddim_sampling_parameters(self.alphas_cumprod, self.ddim_timesteps, ddim_eta)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = sig, a, ap
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1.0 - a)

    # -- one UNet evaluation on static buffers (graph replay when possible) --------------------------------------
    def _prepare(self, x, c_concat, c_fmaps):
        """Static input buffers + (when possible) the captured graph for these shapes and c_fmaps buffers, then the
        conditioning latent of THIS call: c_concat is copied on every sample() — the graph reads the static copy, so a
        caller that refills its tensor in place (or whose new tensor lands on a recycled address) must not meet the
        previous batch's condition.  c_fmaps are read by address and always show their current contents."""
        if self._graph is None:
            dev = x.device
            self._xc = torch.zeros((x.shape[0], x.shape[1] + c_concat.shape[1]) + tuple(x.shape[2:]), device=dev)
            self._t = torch.zeros((x.shape[0],), dtype=torch.long, device=dev)
            self._graph = False
            if self.use_graph and x.is_cuda:
                try:
                    for _ in range(2):
                        self.unet(self._xc, self._t, c_fmaps=c_fmaps)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._out = self.unet(self._xc, self._t, c_fmaps=c_fmaps)
                    self._graph = g
                except Exception:      # capture unsupported: eager launches
                    self._graph = False
        self._xc[:, x.shape[1]:].copy_(c_concat)

    def _eps(self, x, t_int, c_fmaps):
        self._xc[:, :x.shape[1]].copy_(x)
        self._t.fill_(int(t_int))
        if self._graph:
            self._graph.replay()
            return self._out
        return self.unet(self._xc, self._t, c_fmaps=c_fmaps)

    @torch.no_grad()
    def sample(self, S, x_T, c_concat, c_fmaps, eta=1.0, noises=None, generator=None, n_steps=None, temperature=1.0):
        """ddim.py:56-157 (no mask, no guidance, no score corrector — log_images_when_testing uses none).
        x_T (N,4,H,W) start latent, c_concat (N,4,H,W), c_fmaps dict.  Returns (x_0 estimate after the last step,
        {'x_inter': [...], 'pred_x0': [...]}) like the reference; n_steps stops early (timing runs)."""
        self.make_schedule(S, eta)
        key = (tuple(x_T.shape), tuple(c_concat.shape), x_T.device,
               tuple((k, v.data_ptr(), tuple(v.shape)) for k, v in sorted(c_fmaps.items())))
        if key != self._key:      # the captured graph reads the c_fmaps buffers by address: new buffers -> new capture
            self._graph, self._key = None, key
        self._prepare(x_T, c_concat, c_fmaps)
        img = x_T
        total = self.ddim_timesteps.shape[0]
        inter = {"x_inter": [img], "pred_x0": [img]}
        for i, step in enumerate(np.flip(self.ddim_timesteps)):
            if n_steps is not None and i >= n_steps:
                break
            index = total - i - 1
            e_t = self._eps(img, step, c_fmaps)
            if noises is not None:
                nz = noises[i].to(img.device)
            else:
                nz = torch.randn(img.shape, device=img.device, generator=generator)
            img, pred_x0 = self.p_sample_ddim(img, e_t, index, nz * temperature)
            inter["x_inter"].append(img)
            inter["pred_x0"].append(pred_x0)
        return img, inter

    def p_sample_ddim(self, x, e_t, index, noise):
        """ddim.py:159-203 with the model output already evaluated."""
        a_t, a_prev = float(self.ddim_alphas[index]), float(self.ddim_alphas_prev[index])
        sigma_t, sqrt_one_minus_at = float(self.ddim_sigmas[index]), float(self.ddim_sqrt_one_minus_alphas[index])
        pred_x0 = (x - sqrt_one_minus_at * e_t) / (a_t ** 0.5)
        dir_xt = ((1.0 - a_prev - sigma_t ** 2) ** 0.5) * e_t
        return (a_prev ** 0.5) * pred_x0 + dir_xt + sigma_t * noise, pred_x0
