"""Fused concept-slider training iteration on one MI355X (and data-parallel across N of them).

One iteration = the body of the reference loop trainscripts/textsliders/train_lora_xl.py:162-356
(train_lora.py:155-309 for SD-1.x):

    k ~ U{1..max_denoising_steps-1}; latents = randn * init_noise_sigma
    with LoRA on, no grad:   k x [ predict_noise(uncond|target, guidance 3) -> DDIM step ]      (k UNet passes)
    t = timesteps_1000[int(k*1000/50)]                                                           (quirk D.1 kept)
    LoRA off, no grad:       positive / neutral / unconditional predictions, guidance 1         (3 UNet passes)
    LoRA on, grad:           target prediction, guidance 1                                      (1 UNet pass)
    loss = MSE(target, neutral +- gs*(positive - unconditional)); backward; AdamW

Every UNet pass is one command-buffer replay (slh_run_program); CFG combine + DDIM step, the guidance loss,
the backward pass and the flat AdamW are HIP kernels; nothing round-trips through the host inside an
iteration (the reference syncs on loss.item() and empties the allocator cache every iteration).
Data parallel: ranks train different prompt pairs / noise with a shared k, then ONE all-reduce of the flat
fp32 LoRA-gradient buffer (RCCL over xGMI via torch.distributed) before the replicated AdamW step.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import os

import torch

from . import lib
from . import schedulers
from .ddim import DDIMSchedule
from .lora_store import LoraStore
from .parallel import allreduce_sum_, world_info
from .unet import UNetEngine

# train.optimizer values that step the flat parameter buffer with tensor ops (sliders_amd/optim.py) instead of one fused kernel
TENSOR_OP_OPTIMIZERS = ("prodigy", "dadaptadam", "dadaptlion")


@dataclass
class PairEmbeds:
    """Device-resident CFG-concatenated embeddings of one PromptEmbedsPair (prompt_util.py:71-106):
    each ctx tensor is cat([unconditional, X]) as concat_embeddings builds it (train_util.py:136-141)."""
    ctx_target: torch.Tensor          # (2*bs, 77, D)
    ctx_positive: torch.Tensor
    ctx_neutral: torch.Tensor
    ctx_uncond: torch.Tensor
    pooled_target: Optional[torch.Tensor] = None   # (2*bs, P)  SDXL only
    pooled_positive: Optional[torch.Tensor] = None
    pooled_neutral: Optional[torch.Tensor] = None
    pooled_uncond: Optional[torch.Tensor] = None
    guidance_scale: float = 1.0
    action: str = "enhance"


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _ShapeState:
    """Per-(batch, H, W) device buffers of the trainer: prompts may carry their own resolution / batch_size /
    dynamic_resolution (prompt_util.py:44-68, train_lora_xl.py:170-203), so nothing is sized once and for all."""

    def __init__(self, cfg, bs: int, H: int, W: int, dev):
        shape = (bs, cfg.out_channels, H, W)
        z = lambda: torch.zeros(shape, dtype=torch.bfloat16, device=dev)
        self.denoised, self.e_pos, self.e_neu, self.e_unc, self.e_tgt = z(), z(), z(), z(), z()
        self.chw = cfg.out_channels * H * W
        self.time_ids = None
        if cfg.is_xl:   # [orig_h, orig_w, crop_top, crop_left, target_h, target_w], train_util.py:298-333
            self.time_ids = torch.tensor([[H * 8.0, W * 8.0, 0.0, 0.0, H * 8.0, W * 8.0]] * (2 * bs),
                                         dtype=torch.float32, device=dev)


class _StepPrograms:
    """The fused-scheduler denoise loop of one plan, one lib.Program per step index (SliderTrainer._step_programs)."""

    def __init__(self, tr: "SliderTrainer", p_on, key):
        self.key, self.tr, self.p = key, tr, p_on
        B = 2 * tr.bs
        n = len(tr.t50)
        self.t_table = torch.tensor([[float(t)] * B for t in tr.t50], dtype=torch.float32, device=tr.eng.device).reshape(n, B)
        self.progs = [None] * n
        self.captured = False

    def program(self, i: int) -> lib.Program:
        pr = self.progs[i]
        if pr is not None:
            return pr
        tr, p = self.tr, self.p
        # the prompt embeddings do not change inside the loop: after the first step the text K/V are already there
        base = p.prog if (i == 0 or p.prog_text_cached is None) else p.prog_text_cached
        t_io = p.io["t"].ptr
        pr = lib.Program()
        patched = 0
        for (op, d), nm in zip(base.ops, base.op_names):
            if op == lib.OP_TEMBED and d.vals == t_io:
                d = lib.TembedDesc.from_buffer_copy(bytes(d))
                d.vals = self.t_table.data_ptr() + i * self.t_table.shape[1] * 4
                patched += 1
            pr.add(op, d, nm)
        if patched != 1:
            raise RuntimeError(f"step program: {patched} timestep projections read io['t'] (expected one)")
        smp = p.io["sample"]
        half = tr.bs * tr.chw * 2
        pr.add(lib.OP_CFG_DDIM, tr._cfg_desc(p, smp.ptr, tr.denoise_guidance, tr.sched.step_fields(tr.t50[i], tr.nsteps),
                                             out2=smp.ptr + half, x=smp.ptr), "cfg_ddim_step")
        self.progs[i] = pr
        return pr

    def capture_all(self):
        """After the first loop (every kernel of the pass has run): record the graphs of all steps a later k can reach, so
        that no iteration pays a capture in its denoise loop."""
        if self.captured:
            return
        self.captured = True
        for i in range(max(1, self.tr.nsteps - 1)):
            if not self.program(i).capture():
                break


class SliderTrainer:
    def __init__(self, engine: UNetEngine, store: LoraStore, H: int, W: int, batch_size: int = 1,
                 lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                 max_denoising_steps: int = 50, denoise_guidance: float = 3.0, process_group=None,
                 dedup_frozen: bool = True, prediction_type: str = "epsilon", optimizer: str = "adamw",
                 noise_scheduler: str = "ddim", scheduler_seed: int = 0, optimizer_kwargs: Optional[dict] = None):
        self.eng, self.store = engine, store
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        if optimizer not in ("adamw", "adam", "lion") + TENSOR_OP_OPTIMIZERS:
            raise NotImplementedError(f"optimizer '{optimizer}': fused flat kernels exist for adam / adamw (slh_adamw) and lion "
                                      f"(slh_lion); prodigy / dadaptadam / dadaptlion run as sliders_amd.optim classes on the flat "
                                      f"parameter buffer")
        if getattr(store, "master", None) is not None and optimizer not in ("adamw", "adam"):
            raise NotImplementedError(f"optimizer '{optimizer}' with fp32 adapter state (train.precision: float32): only the fused "
                                      f"adam / adamw kernel keeps an fp32 master; use bfloat16 for lion / prodigy / dadapt*")
        self.optimizer = optimizer
        self.optimizer_kwargs = dict(optimizer_kwargs or {})
        self._tensor_opt = None
        self.nsteps = max_denoising_steps
        self.denoise_guidance = denoise_guidance
        # train.noise_scheduler (model_util.py:230-277); v_prediction: pretrained_model.v_pred (model_util.py:126)
        if noise_scheduler.lower().replace(" ", "_") == "ddim":
            self.sched = DDIMSchedule(prediction_type=prediction_type)
        else:
            self.sched = schedulers.create(noise_scheduler, prediction_type)
            # ddpm / euler_a draw device noise in every step (the reference uses the global device RNG there)
            self.sched_generator = torch.Generator(device=engine.device)
            self.sched_generator.manual_seed(int(scheduler_seed))
        self.pg = process_group
        self.rank, self.world = world_info(process_group)
        self.time_allreduce = False          # bench.py: bracket every gradient all-reduce with events (reduce_and_step)
        self._ar_events = []
        self.grad_scale = 1.0
        engine.attach_lora(store) if engine.lora is not store else None
        self.loss = torch.zeros(1, dtype=torch.float32, device=engine.device)
        if self.sched.fused:
            self.t50 = self.sched.make_timesteps(max_denoising_steps)
            self.t1000 = self.sched.make_timesteps(1000)
        self.unet_passes = 0
        self.dedup_frozen = dedup_frozen
        # the three frozen predictions and the training forward only share their input (the denoised latents): they run on two
        # streams (+1 % on the SDXL bench: 42.16 -> 42.58 steps/s in one call; the whole trainer / parity / seam / RCCL test set
        # passes either way, including the bit-equality checks).  SLIDERS_OVERLAP_FROZEN=0 runs them back to back.
        self.overlap_frozen = os.environ.get("SLIDERS_OVERLAP_FROZEN", "1") == "1"
        self._side = torch.cuda.Stream(device=engine.device) if self.overlap_frozen else None
        # one program (one hipGraph) per denoise step instead of fill + pass + combine launch; SLIDERS_STEP_GRAPHS=0: the latter
        self.step_graphs = os.environ.get("SLIDERS_STEP_GRAPHS", "1") == "1"
        # steps of the denoise loop left when the side stream's work is queued (0 = as soon as the host gets there)
        self.frozen_gate = int(os.environ.get("SLIDERS_FROZEN_GATE", "1"))
        self._captured = set()
        self.phase_events = None
        self.phase_steps = False
        self._states = {}
        self._use(batch_size, H, W)

    def _use(self, bs: int, H: int, W: int):
        """Select (and lazily create) the buffers of one (batch, H, W); `iteration` calls it from the noise shape."""
        key = (bs, H, W)
        st = self._states.get(key)
        if st is None:
            st = self._states[key] = _ShapeState(self.eng.cfg, bs, H, W, self.eng.device)
        self.H, self.W, self.bs = H, W, bs
        self.denoised, self.e_pos, self.e_neu, self.e_unc, self.e_tgt = st.denoised, st.e_pos, st.e_neu, st.e_unc, st.e_tgt
        self.chw, self.time_ids = st.chw, st.time_ids

    # ---- helpers ------------------------------------------------------------------------------------
    def _load_cond(self, p, ctx, pooled):
        p.io["ctx"].tensor.copy_(ctx)
        if self.eng.cfg.is_xl:
            p.io["time_ids"].tensor.copy_(self.time_ids)
            p.io["add_in"].tensor[:, : self.eng.cfg.pooled_dim].copy_(pooled)

    def _load_latents(self, p, lat):
        s = p.io["sample"].tensor
        s[: self.bs].copy_(lat)
        s[self.bs:].copy_(lat)

    def _model_input(self, lat, t):
        """scheduler.scale_model_input (train_util.py:156, 234): identity for ddim / ddpm, 1/sqrt(sigma^2+1) for lms / euler_a"""
        return lat if self.sched.fused else self.sched.scale_model_input(lat, t)

    def _denoise_unfused(self, p_on, noise, k: int):
        """train_util.diffusion[_xl] (train_util.py:175-196 / 263-294) for the tensor-op schedulers: UNet pass and guidance
        combine are the HIP path, the scheduler step is sliders_amd/schedulers.py on the device latents.  Leaves the
        scheduler on its 1000-step grid like train_lora_xl.py:229 and returns the timestep of the four predictions."""
        sch, s = self.sched, _stream()
        eps = self.e_tgt                    # scratch: the guided prediction of this step (overwritten again in step 3)
        first = [True]

        def predict(model_input, t):
            self._load_latents(p_on, model_input)
            p_on.io["t"].tensor.fill_(float(t))
            # the prompt embeddings do not change inside the loop: after the first step the text K/V are already there
            (p_on.prog if (first[0] or p_on.prog_text_cached is None) else p_on.prog_text_cached).run(s)
            first[0] = False
            self.unet_passes += 1
            self._cfg(p_on, eps.data_ptr(), self.denoise_guidance)
            return eps

        lat = schedulers.denoise(sch, predict, noise.to(device=self.eng.device, dtype=torch.bfloat16), k, self.nsteps,
                                 generator=self.sched_generator, device=self.eng.device)
        self.denoised.copy_(lat)
        sch.set_timesteps(1000, device=self.eng.device)
        return sch.timesteps[int(k * 1000 / self.nsteps)]

    def _cfg_desc(self, p, out, guidance, coeff=None, out2=None, x=None, eps_text=None):
        d = lib.CfgDdimDesc(eps=p.io["eps"].ptr, x=x or 0, out=out, out2=out2 or 0, eps_text=eps_text or 0,
                            nb=self.bs, chw=self.chw, guidance=guidance, do_step=0)
        if coeff is not None:
            for kf, vf in coeff.items():
                setattr(d, kf, vf)
        return d

    def _cfg(self, p, out, guidance, coeff=None, out2=None, x=None, eps_text=None):
        lib.call(lib.OP_CFG_DDIM, self._cfg_desc(p, out, guidance, coeff, out2, x, eps_text), _stream())

    def _step_programs(self, p_on) -> "_StepPrograms":
        """Denoise step i as ONE replayable program: the pass reading its timestep from row i of a device table instead of
        io["t"], followed by the guided combine + DDIM update with step i's coefficients (train_util.py:263-294 per
        step: scale_model_input, predict_noise_xl, scheduler.step).  The loop then submits one graph per step - no fill
        kernel and no separate launch between two passes (scripts/time_iter_k.py: 0.32 ms per step of the SDXL bench)."""
        key = (id(self), self.bs, self.chw, self.nsteps, float(self.denoise_guidance))
        sp = getattr(p_on, "_step_progs", None)
        if sp is None or sp.key != key:
            sp = p_on._step_progs = _StepPrograms(self, p_on, key)
        return sp

    def _predict(self, p, lat, ctx, pooled, t, out):
        self._load_latents(p, self._model_input(lat, t))
        self._load_cond(p, ctx, pooled)
        p.io["t"].tensor.fill_(float(t))
        p.prog.run(_stream())
        self.unet_passes += 1
        self._cfg(p, out.data_ptr(), 1.0)

    def _frozen_dedup(self, pair: PairEmbeds, t_cur: int):
        """The reference's three frozen predictions are CFG pairs [uncond; positive], [uncond; neutral] and
        [uncond; uncond] on the SAME latents and timestep (train_lora_xl.py:236-295): five of the six samples
        are the unconditional one.  One UNet pass over [uncond, positive, neutral] produces the same three epsilon
        tensors (kernels are deterministic per sample); the combines `u + 1*(x - u)` keep the reference's bf16
        rounding.  Still counted as 3 denoise steps (that is what the reference executes)."""
        eng, bs = self.eng, self.bs
        p3 = eng.plan(3 * bs, self.H, self.W, "off")
        s = p3.io["sample"].tensor
        lat_in = self._model_input(self.denoised, t_cur)
        for j in range(3):
            s[j * bs:(j + 1) * bs].copy_(lat_in)
        ctx = p3.io["ctx"].tensor
        ctx[:bs].copy_(pair.ctx_uncond[:bs]); ctx[bs:2 * bs].copy_(pair.ctx_positive[bs:]); ctx[2 * bs:].copy_(pair.ctx_neutral[bs:])
        if eng.cfg.is_xl:
            p3.io["time_ids"].tensor.copy_(self.time_ids[:1].expand(3 * bs, 6))
            ai = p3.io["add_in"].tensor
            pd = eng.cfg.pooled_dim
            ai[:bs, :pd].copy_(pair.pooled_uncond[:bs]); ai[bs:2 * bs, :pd].copy_(pair.pooled_positive[bs:])
            ai[2 * bs:, :pd].copy_(pair.pooled_neutral[bs:])
        p3.io["t"].tensor.fill_(float(t_cur))
        p3.prog.run(_stream())
        self.unet_passes += 3
        e = p3.io["eps"].ptr
        blk = bs * self.chw * 2
        self._cfg(p3, self.e_pos.data_ptr(), 1.0, eps_text=e + blk)
        self._cfg(p3, self.e_neu.data_ptr(), 1.0, eps_text=e + 2 * blk)
        self._cfg(p3, self.e_unc.data_ptr(), 1.0, eps_text=e)

    # ---- one iteration ------------------------------------------------------------------------------
    def iteration(self, pair: PairEmbeds, k: int, noise: torch.Tensor, lr: Optional[float] = None,
                  time_ids: Optional[torch.Tensor] = None, zero_grads: bool = True, step: bool = True) -> torch.Tensor:
        """noise: (bs,4,H,W) already scaled by the scheduler's init_noise_sigma (train_util.py:55; 1 for ddim / ddpm, the
        largest sigma for lms / euler_a: `trainer.sched.init_noise_sigma`); its shape selects the resolution and batch of
        this iteration.  lr: this step's learning rate (the host evaluates the LR schedule, train_lora_xl.py:346-347).
        time_ids: (2*bs, 6) SDXL micro-conditioning when it is not the default [H,W,0,0,H,W] (dynamic_crops).
        zero_grads = False: this pair's gradient ADDS to what the flat buffer holds (gradient accumulation over pairs);
        step = False: stop after the backward - no all-reduce, no optimizer step (the caller finishes with `reduce_and_step`;
        this is also how tests run N data-parallel ranks in one process).
        Returns the device loss scalar."""
        self._use(noise.shape[0], noise.shape[2], noise.shape[3])
        self._mark("start")
        if time_ids is not None:
            self.time_ids = time_ids.to(device=self.eng.device, dtype=torch.float32).reshape(2 * self.bs, 6)
        if lr is not None:
            self.lr = float(lr)
        eng, st, bs = self.eng, self.store, self.bs
        B = 2 * bs
        s = _stream()
        gate = None
        # 1. partial denoise with the adapters on (train_lora_xl.py:205-227)
        eng.set_lora(True, 1.0)
        p_on = eng.plan(B, self.H, self.W, "on")
        self._load_cond(p_on, pair.ctx_target, pair.pooled_target)
        if self.sched.fused:
            self._load_latents(p_on, noise.to(torch.bfloat16))
            smp = p_on.io["sample"]
            half = bs * self.chw * 2
            sp = self._step_programs(p_on) if self.step_graphs else None
            # (the event sits in front of step k - frozen_gate: it fires when exactly `frozen_gate` steps are left)
            gate_at = k - self.frozen_gate if (self.overlap_frozen and self.frozen_gate > 0) else -1
            for i in range(k):
                self.unet_passes += 1
                if i == gate_at:
                    gate = torch.cuda.Event()
                    gate.record(torch.cuda.current_stream())
                if sp is not None:
                    sp.program(i).run(s)
                    if self.phase_events is not None and self.phase_steps:
                        self._mark(f"step{i}")
                    continue
                t = self.t50[i]
                p_on.io["t"].tensor.fill_(float(t))
                # the prompt embeddings do not change inside the loop: after the first step the text K/V are already there
                (p_on.prog if (i == 0 or p_on.prog_text_cached is None) else p_on.prog_text_cached).run(s)
                self._cfg(p_on, smp.ptr, self.denoise_guidance, self.sched.step_fields(t, self.nsteps),
                          out2=smp.ptr + half, x=smp.ptr)
            if sp is not None:
                sp.capture_all()
            self._mark("denoise")
            self.denoised.copy_(smp.tensor[:bs])
            t_cur = self.t1000[int(k * 1000 / self.nsteps)]
        else:
            t_cur = self._denoise_unfused(p_on, noise, k)
        # 2. frozen-model predictions, adapters off (train_lora_xl.py:236-295)
        def frozen():
            if self.dedup_frozen:
                self._frozen_dedup(pair, t_cur)
            else:
                p_off = eng.plan(B, self.H, self.W, "off")
                self._predict(p_off, self.denoised, pair.ctx_positive, pair.pooled_positive, t_cur, self.e_pos)
                self._predict(p_off, self.denoised, pair.ctx_neutral, pair.pooled_neutral, t_cur, self.e_neu)
                self._predict(p_off, self.denoised, pair.ctx_uncond, pair.pooled_uncond, t_cur, self.e_unc)

        eng.set_lora(False)
        frozen_done = None
        if self.overlap_frozen:
            # adapter-free plans have their own arena and launch no adapter work (they never read the scale the main
            # stream is about to switch back on), so they can run beside step 3
            eng.plan(3 * bs if self.dedup_frozen else B, self.H, self.W, "off")      # built (and arenas sized) on the main stream
            eng.plan(B, self.H, self.W, "train")
            main = torch.cuda.current_stream()
            if gate is not None:
                # The host runs ~25 steps ahead of the device; work queued on the side stream THEN sits in a second hardware queue behind
                # its cross-queue wait for the rest of the loop, and every step of the loop runs 2 % slower while it does (SDXL 1024^2:
                # 22.36 -> 22.88 ms from the step at which the host got there; profiles/r06_iteration_k_sweep.txt) - about what the
                # overlap gains.  So the host holds the submission back until the loop has `frozen_gate` steps left (a wait on an event,
                # no data comes back); those steps are more than the time it takes to queue the two passes.
                gate.synchronize()
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                frozen()
                frozen_done = torch.cuda.Event()
                frozen_done.record(self._side)
        else:
            frozen()
        # 3. target prediction with the adapters on, kept for backward (train_lora_xl.py:302-322)
        eng.set_lora(True, 1.0)
        p_tr = eng.plan(B, self.H, self.W, "train")
        self._predict(p_tr, self.denoised, pair.ctx_target, pair.pooled_target, t_cur, self.e_tgt)
        if frozen_done is not None:
            torch.cuda.current_stream().wait_event(frozen_done)
        self._mark("predictions")
        # 4. loss + its gradient, written straight into the backward plan's input (prompt_util.py:108-148)
        self.loss.zero_()
        bw = p_tr.backward
        n = bs * self.chw
        d = lib.LossDesc(target=self.e_tgt.data_ptr(), positive=self.e_pos.data_ptr(), neutral=self.e_neu.data_ptr(),
                         uncond=self.e_unc.data_ptr(), loss=self.loss.data_ptr(), dtarget=0,
                         dtarget_pix=bw.deps_pix.ptr, n=n, guidance=float(pair.guidance_scale),
                         erase=1 if pair.action == "erase" else 0, hw=self.H * self.W, nch=eng.cfg.out_channels)
        lib.call(lib.OP_LOSS, d, s)
        # 5. backward into the flat fp32 gradient buffer, (all-reduce,) AdamW
        if zero_grads:
            st.grads.zero_()
        bw.prog.run(s)
        self._mark("backward")
        if step:
            self.reduce_and_step()
            self._mark("optimizer")
        if self.step_graphs:
            # the once-per-iteration programs of this shape (frozen pass, training forward, backward): recorded as graphs behind their
            # FIRST run - every kernel of theirs has been launched by then - instead of inside the third iteration (lib.Program.run's
            # own rule), so that a run with one warm-up iteration has every capture behind it when its timing starts
            key = (B, self.H, self.W, self.dedup_frozen)
            if key not in self._captured:
                self._captured.add(key)
                off = eng.plan(3 * bs if self.dedup_frozen else B, self.H, self.W, "off")
                for prog in (off.prog, p_tr.prog, bw.prog):
                    if prog._runs >= 1:
                        prog.capture()
        return self.loss

    def _mark(self, name: str):
        """Phase stamps for scripts/time_iter_k.py: with `phase_events` set to a list, an event per phase boundary on the main stream."""
        if self.phase_events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            self.phase_events.append((name, e))

    def reduce_and_step(self, n_accumulated: int = 1):
        """The exchange step of data parallelism and the optimizer: ONE sum all-reduce of the flat fp32 gradient buffer per
        optimizer step, issued on the stream the backward ran on (the collective is ordered behind the last weight-gradient
        launch and ahead of the optimizer kernel by stream order alone - no host synchronisation, no side stream), the mean over
        ranks (x accumulated pairs) folded into the optimizer kernel.  With `time_allreduce` set the collective is bracketed by
        two events on that stream; `allreduce_ms()` reads them back after the caller's synchronisation."""
        st = self.store
        cur = torch.cuda.current_stream()
        timed = self.time_allreduce and (self.pg is not None or self.world > 1)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        self.grad_scale = allreduce_sum_(st.grads, self.pg) / float(n_accumulated)
        if timed:
            e1.record(cur)
            self._ar_events.append((e0, e1))
        self.optimizer_step()

    def allreduce_ms(self):
        """Event times of the all-reduces since the last call (ms each); call after a device synchronisation."""
        out = [a.elapsed_time(b) for a, b in self._ar_events]
        self._ar_events = []
        return out

    def optimizer_step(self):
        st = self.store
        st.opt_step += 1
        if self.optimizer in TENSOR_OP_OPTIMIZERS:
            # prodigyopt 1.0 / dadaptation 3.1 over the flat bf16 parameter buffer (train_util.py:339-346, 369-372)
            if self._tensor_opt is None:
                from . import optim
                cls = {"prodigy": optim.Prodigy, "dadaptadam": optim.DAdaptAdam, "dadaptlion": optim.DAdaptLion}[self.optimizer]
                kw = {k: v for k, v in self.optimizer_kwargs.items()}
                if self.eps and self.optimizer != "dadaptlion":
                    kw.setdefault("eps", self.eps)
                self._tensor_opt = cls([st.params], lr=self.lr, betas=self.betas, weight_decay=self.wd, **kw)
            self._tensor_opt.param_groups[0]["lr"] = self.lr
            # the reference's gradients have the parameter dtype (bf16); grad_scale carries the 1/world of the all-reduce
            st.params.grad = (st.grads * self.grad_scale).to(st.params.dtype)
            self._tensor_opt.step()
            st.params.grad = None
            return
        if self.optimizer == "lion":        # one moment (store.exp_avg); lion_pytorch 0.1.2 op order
            d = lib.LionDesc(param=st.params.data_ptr(), exp_avg=st.exp_avg.data_ptr(), grad=st.grads.data_ptr(), n=st.numel,
                             lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], weight_decay=self.wd,
                             grad_scale=self.grad_scale)
            lib.call(lib.OP_LION, d, _stream())
            return
        f32 = st.master is not None          # train.precision float32: fp32 master + moments, bf16 copy rewritten in the same launch
        d = lib.AdamwDesc(param=(st.master if f32 else st.params).data_ptr(), exp_avg=st.exp_avg.data_ptr(),
                          exp_avg_sq=st.exp_avg_sq.data_ptr(), grad=st.grads.data_ptr(), n=st.numel, lr=self.lr, beta1=self.betas[0],
                          beta2=self.betas[1], eps=self.eps, weight_decay=self.wd, step=st.opt_step, grad_scale=self.grad_scale,
                          param_lo=st.params.data_ptr() if f32 else 0, f32_state=1 if f32 else 0)
        lib.call(lib.OP_ADAMW, d, _stream())
