"""`unet(sample, t, encoder_hidden_states, added_cond_kwargs)['sample']` seam
(models/unet_2d_condition.py:703-717) backed by the HIP engine."""
import types

import torch

from .engine import Engine


class HipUNet2DConditionModel:
    """Holds the engine(s) for one set of weights; engines are created lazily per latent size."""

    def __init__(self, config, state_dict=None, device=0, max_streams=8, max_prompts=8):
        self.config_dict = dict(config)
        self.config = types.SimpleNamespace(**config)
        self.in_channels = config["in_channels"]
        self.device_index = device
        self.max_streams, self.max_prompts = max_streams, max_prompts
        self._state_dict = state_dict
        self._engines = {}

    def load_state_dict(self, sd):
        self._state_dict = sd
        for e in self._engines.values():
            e.load_state_dict(sd)

    def engine(self, h, w, streams=None, prompts=None):
        """Engine for latent size (h, w).  `streams` / `prompts`: what the caller is about to use (F batched forwards per step,
        P prompts in the K/V cache); an engine sized for fewer is rebuilt once with room for them (<= 16 streams)."""
        grow = False
        if streams and streams > self.max_streams:
            if streams > 16:
                raise ValueError(f"{streams} batched UNet forwards per step exceed the engine limit of 16 (at most 13 regions with injection)")
            self.max_streams, grow = min(16, max(streams, 2 * self.max_streams)), True
        if prompts and prompts > self.max_prompts:
            self.max_prompts, grow = max(prompts, 2 * self.max_prompts), True
        empty = isinstance(self._state_dict, str) and self._state_dict == "empty"
        # A model built "empty" (ranks != 0 of a seed-parallel launch) holds weights that exist nowhere else on this rank: the packed
        # arena it received by launcher.broadcast_weights.  Its layout is a function of the config alone (not of the stream / prompt
        # capacity or the latent size), so a new engine - a grown one, or one for another latent size - takes it over device to
        # device from ANY engine that is bound.  The donor stays registered until the copy has succeeded: if the
        # allocation of the new engine fails, nothing is lost.
        donor = next((e for e in self._engines.values() if e.weights_missing()[0] == 0), None) if empty else None
        stale = dict(self._engines) if grow else {}
        if grow:
            self._engines = {}
        key = (h, w)
        if key not in self._engines:
            if self._state_dict is None:
                raise RuntimeError("HipUNet2DConditionModel: no weights loaded (call load_state_dict)")
            try:
                e = Engine(self.config_dict, h, w, device=self.device_index, max_streams=self.max_streams,
                           max_prompts=self.max_prompts)
                try:
                    if empty:
                        if donor is not None:
                            from .launcher import arena_tensor
                            src, dst = arena_tensor(donor), arena_tensor(e)
                            if src.numel() != dst.numel():
                                raise RuntimeError(f"weight arena layout changed with the engine capacity ({src.numel()} vs {dst.numel()} bytes)")
                            dst.copy_(src)
                            torch.cuda.synchronize(self.device_index)
                            e.arena_mark_bound()
                        # otherwise the packed arena arrives by launcher.broadcast_weights
                    elif isinstance(self._state_dict, str) and self._state_dict.startswith("random"):
                        e.init_random_weights(seed=int(self._state_dict[6:] or 0))      # "random<seed>": benchmarks without checkpoints
                    else:
                        e.load_state_dict(self._state_dict)
                except BaseException:
                    e.close()
                    raise
            except BaseException:
                if grow:                                   # the old engines (and the only copy of a broadcast-fed rank's weights) stay usable
                    self._engines = stale
                raise
            self._engines[key] = e
        for old in stale.values():                         # the rebuilt engine is bound: the smaller ones can go
            old.close()
        return self._engines[key]

    def __call__(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, **kw):
        """Batch of independent samples; every sample uses its own row of `encoder_hidden_states`."""
        B, _, h, w = sample.shape
        eng = self.engine(h, w)
        pooled = tid = None
        if added_cond_kwargs is not None:
            pooled, tid = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"][:1]
        eng.set_prompts(encoder_hidden_states, pooled, tid)
        eng.set_fontsize(None, None)
        out = eng.unet_forward(sample, float(timestep), list(range(B)))
        return {"sample": out}
