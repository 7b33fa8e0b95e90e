"""TEST INFRASTRUCTURE (CPU oracle) - never imported by the product path.

float64 / numpy restatement of the three non-default noise schedulers the reference can be configured with
(trainscripts/textsliders/model_util.py:247-274: "ddpm", "lms", "euler_a"; consumed at train_util.py:55 init_noise_sigma,
train_util.py:156/234 scale_model_input, train_util.py:193/291 step(...).prev_sample).  The algorithms live in the
third-party dependency diffusers (requirements: diffusers==0.20.2), which is absent from /root/reference and from this
image: **parity unpinned** - what is restated here is the published algorithm of

  DDPMScheduler.step                    (Ho et al. 2020, eq. 7; variance_type "fixed_small", clip_sample False)
  EulerAncestralDiscreteScheduler.step  (k-diffusion sample_euler_ancestral, eta = 1)
  LMSDiscreteScheduler.step             (k-diffusion sample_lms, order 4)

with the configuration the reference passes (scaled_linear betas 0.00085 .. 0.012, 1000 train steps).  Independent of
the product code on purpose: closed forms in float64 (the product mirrors the library's tensor-op order in the working
dtype), the LMS coefficients by exact polynomial integration (the library and the product use scipy.integrate.quad).
External pins checked by tests/test_schedulers.py: sigma_max = 14.6146, sigma_min = 0.0292 (the k-diffusion constants
for this beta schedule).
"""
import numpy as np

T = 1000


def alphas_cumprod(beta_start=0.00085, beta_end=0.012):
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas)


def sigmas_train():
    ac = alphas_cumprod()
    return np.sqrt((1.0 - ac) / ac)


# ---- DDPM -------------------------------------------------------------------------------------------------------
def ddpm_timesteps(n):
    """timestep_spacing "leading": the same grid as DDIM"""
    return (np.arange(n) * (T // n))[::-1].copy()


def ddpm_step(x, model_out, t, n_steps, noise, prediction_type="epsilon"):
    ac = alphas_cumprod()
    prev_t = t - T // n_steps
    a_t = ac[t]
    a_p = ac[prev_t] if prev_t >= 0 else 1.0
    cur_alpha = a_t / a_p
    cur_beta = 1.0 - cur_alpha
    if prediction_type == "epsilon":
        x0 = (x - np.sqrt(1 - a_t) * model_out) / np.sqrt(a_t)
    else:
        x0 = np.sqrt(a_t) * x - np.sqrt(1 - a_t) * model_out
    mean = (np.sqrt(a_p) * cur_beta / (1 - a_t)) * x0 + (np.sqrt(cur_alpha) * (1 - a_p) / (1 - a_t)) * x
    if t == 0:
        return mean
    var = max((1 - a_p) / (1 - a_t) * cur_beta, 1e-20)
    return mean + np.sqrt(var) * noise


# ---- sigma-space schedulers (Euler ancestral, LMS) -----------------------------------------------------------------
def sigma_timesteps(n):
    """timestep_spacing "linspace": float timesteps"""
    return np.linspace(0, T - 1, n, dtype=np.float64)[::-1].copy()


def sigma_schedule(n):
    """sigmas at the n inference timesteps (linear interpolation of the training sigmas) followed by 0"""
    s = np.interp(sigma_timesteps(n), np.arange(T), sigmas_train())
    return np.concatenate([s, [0.0]])


def scale_model_input(x, i, n):
    s = sigma_schedule(n)[i]
    return x / np.sqrt(s * s + 1.0)


def _x0(x, model_out, s, prediction_type):
    if prediction_type == "epsilon":
        return x - s * model_out
    return model_out * (-s / np.sqrt(s * s + 1)) + x / (s * s + 1)


def euler_a_step(x, model_out, i, n, noise, prediction_type="epsilon"):
    sig = sigma_schedule(n)
    s, s_to = sig[i], sig[i + 1]
    x0 = _x0(x, model_out, s, prediction_type)
    s_up = np.sqrt(s_to ** 2 * (s ** 2 - s_to ** 2) / s ** 2)
    s_down = np.sqrt(s_to ** 2 - s_up ** 2)
    d = (x - x0) / s
    return x + d * (s_down - s) + noise * s_up


def lms_coefficients(sig, i, order):
    """integral over [sig[i], sig[i+1]] of the Lagrange basis polynomials through sig[i], sig[i-1], ..; exact"""
    out = []
    for j in range(order):
        p = np.poly1d([1.0])
        for k in range(order):
            if k == j:
                continue
            p = p * np.poly1d([1.0, -sig[i - k]]) / (sig[i - j] - sig[i - k])
        P = p.integ()
        out.append(P(sig[i + 1]) - P(sig[i]))
    return out


def lms_run(x, model_outs, n, prediction_type="epsilon", order=4):
    """apply len(model_outs) LMS steps starting at index 0 with GIVEN model outputs; returns the list of states"""
    sig = sigma_schedule(n)
    derivs, xs = [], []
    for i, m in enumerate(model_outs):
        s = sig[i]
        d = (x - _x0(x, m, s, prediction_type)) / s
        derivs.append(d)
        derivs = derivs[-order:]
        o = min(i + 1, order)
        c = lms_coefficients(sig, i, o)
        x = x + sum(cj * dj for cj, dj in zip(c, reversed(derivs)))
        xs.append(x)
    return xs


# ---- Euler (deterministic), "leading" spacing with steps_offset: the SDXL checkpoints' scheduler_config -----------
def leading_timesteps(n, offset=1):
    return (np.arange(n) * (T // n))[::-1].astype(np.float64) + offset


def leading_sigma_schedule(n, offset=1):
    s = np.interp(leading_timesteps(n, offset), np.arange(T), sigmas_train())
    return np.concatenate([s, [0.0]])


def euler_step(x, model_out, i, sig, prediction_type="epsilon"):
    """k-diffusion sample_euler with s_churn = 0: x + d * (sigma_next - sigma), d = (x - x0) / sigma"""
    s = sig[i]
    return x + (x - _x0(x, model_out, s, prediction_type)) / s * (sig[i + 1] - s)
