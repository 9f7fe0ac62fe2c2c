"""Host mirrors of a quantiser's tiny device buffers (`bit`, `has_inited_quant_para`).

The reference reads these buffers on every forward (`if self.has_inited_quant_para == 0`, `self.bit > 6`:
AQ:470, :482), which costs a device->host sync each time.  Here the host keeps the last value it read or wrote,
validated by (data_ptr, _version, device) of the buffer: an in-place edit (`fill_`, `copy_`, load_state_dict), a
rebinding (`.bit.data = torch.tensor(8)`, AQ/quant_model.py:83) or nothing at all are told apart without touching
the device; only a changed buffer is read back, once.  Device / dtype moves (`.to()`, `.cuda()`, `.half()`) keep
the value and re-key the mirror.

What the key cannot see: a write through `.data` that lands on the SAME address with the same `_version` (an in-place
`buf.data.copy_(...)`, or two rebindings between forwards whose second allocation recycles the first address).  The
in-package writers (`_to_8bit`, `set_8_bit_layer_*`, `load_state_dict`, calibration) re-key or re-arm themselves; external
code that edits `bit` / `has_inited_quant_para` / `quant_grid` that way must call the quantiser's `rearm()` afterwards
(INTEGRATION.md, section 2).  The operator-level drop-in (`quant_cuda.quant`) does not rely on keys at all: its kernel
verifies the grid on the device.
"""


class HostMirrorMixin:
    def _hm_setup(self, **values):
        self._hm = {name: [self._hm_key(name), value] for name, value in values.items()}

    def _hm_key(self, name):
        t = getattr(self, name)
        return (t.data_ptr(), t._version, t.device)

    def _hm_get(self, name):
        ent = self._hm[name]
        key = self._hm_key(name)
        if ent[0] != key:
            ent[1] = getattr(self, name).item()      # the buffer changed behind our back: one read-back
            ent[0] = key
        return ent[1]

    def _hm_known(self, name, value):
        """The host just wrote `value` into the buffer itself."""
        self._hm[name] = [self._hm_key(name), value]

    def _hm_fresh(self):
        return all(ent[0] == self._hm_key(name) for name, ent in self._hm.items())

    def _apply(self, fn, *args, **kwargs):
        fresh = {name: ent[0] == self._hm_key(name) for name, ent in self._hm.items()}
        out = super()._apply(fn, *args, **kwargs)
        for name, ok in fresh.items():
            if ok:
                self._hm[name][0] = self._hm_key(name)
        return out
