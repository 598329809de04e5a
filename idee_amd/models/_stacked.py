"""Helpers for modules that keep the V per-variable weight sets as one stacked parameter but
expose / accept the reference's per-variable state-dict keys."""
from __future__ import annotations

import torch
import torch.nn as nn


class StackedModule(nn.Module):
    """Subclasses implement `_key_map()` -> list of (reference_key, tensor_name, index | None):
    reference_key (relative to this module) holds `getattr(self, tensor_name)[index]` (or the
    whole tensor when index is None).  state_dict()/load_state_dict() then speak the reference's
    keys; the stacked tensors themselves never appear."""

    def _key_map(self):
        raise NotImplementedError

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for key, name, idx in self._key_map():
            t = getattr(self, name)
            t = t if idx is None else t[idx]
            destination[prefix + key] = t if keep_vars else t.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        expected = set()
        with torch.no_grad():
            for key, name, idx in self._key_map():
                full = prefix + key
                expected.add(full)
                if full not in state_dict:
                    if strict:
                        missing_keys.append(full)
                    continue
                src = state_dict[full]
                dst = getattr(self, name)
                dst = dst if idx is None else dst[idx]
                if tuple(src.shape) != tuple(dst.shape):
                    error_msgs.append(f"size mismatch for {full}: checkpoint {tuple(src.shape)} vs model {tuple(dst.shape)}")
                    continue
                dst.copy_(src)
        if strict:
            own_children = tuple(prefix + n + "." for n, _ in self.named_children())
            for k in state_dict:
                if k.startswith(prefix) and k not in expected and not k.startswith(own_children):
                    unexpected_keys.append(k)


def reference_param_slices(model):
    """[(reference key, stacked Parameter, index | None, reference shape)] in the order of the REFERENCE model's `model.parameters()` — the
    index space of torch.optim.Adam's state dict in a reference checkpoint (utils_train.py:562-584).  The reference registers
    one small parameter per variable where this code base keeps one V-stacked parameter; entry i is the slice
    `parameter[index]` (or the whole parameter) that the reference's i-th parameter corresponds to.  Walks the module tree in
    registration order (= the reference's, verified against the golden state-dict key order); buffers are skipped."""
    out = []
    for prefix, mod in model.named_modules():
        pre = prefix + "." if prefix else ""
        if isinstance(mod, StackedModule):
            for key, name, idx in mod._key_map():
                p = mod._parameters.get(name)
                if p is None and name.endswith("_w5"):        # "<x>_w5": the (…,1,1,1) state-dict view of parameter "<x>_w"
                    p = mod._parameters.get(name[:-1])
                if p is None:
                    continue                                   # a buffer (relative_position_index)
                t = getattr(mod, name)                        # the state-dict-shaped view (e.g. (V, C, Cin, 1, 1, 1))
                out.append((pre + key, p, idx, tuple((t if idx is None else t[idx]).shape)))
        else:
            for name, p in mod._parameters.items():
                if p is not None:
                    out.append((pre + name, p, None, tuple(p.shape)))
    return out
