"""Reader of the model checkpoints the reference's training writes (Lightning ``.ckpt``).

A Lightning checkpoint is a ``torch.save`` zip archive holding a dict with ``state_dict`` (the
tensors, keys prefixed ``generator_network.`` / ``var_latent_network.``, irwg/models/vae.py:127-128)
next to ``hyper_parameters``, optimizer states and callback states.  The reference's VAE stores its
two networks and its distribution enums in ``hyper_parameters`` (``save_hyperparameters()`` of a
constructor that takes ``nn.Module`` arguments, vae.py:99-122), so the archive's pickle stream names
classes of the training code and ``torch.load(weights_only=True)`` refuses it.

``load_state_dict_from_checkpoint`` never executes anything the file names: the pickle stream is
read by an unpickler whose ``find_class`` resolves a short allow-list of tensor-rebuilding
functions and replaces every other global by an inert placeholder (construction, ``__setstate__``
and calls do nothing), and whose ``persistent_load`` reads the tensor storages from the archive's
``data/<key>`` records.  Only ``state_dict`` is returned.
"""
import collections
import pickle
import zipfile

import torch

_DTYPES = {
    "FloatStorage": torch.float32, "DoubleStorage": torch.float64, "HalfStorage": torch.float16,
    "BFloat16Storage": torch.bfloat16, "LongStorage": torch.int64, "IntStorage": torch.int32,
    "ShortStorage": torch.int16, "CharStorage": torch.int8, "ByteStorage": torch.uint8, "BoolStorage": torch.bool,
}


class _StorageType:
    """Stands for ``torch.<X>Storage`` in a persistent id."""

    def __init__(self, name):
        self.dtype = _DTYPES[name]


class _Inert:
    """Placeholder for every global that is not on the allow-list."""

    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, *args, **kwargs):
        return _Inert()

    def __setstate__(self, state):
        pass

    def __setitem__(self, key, value):
        pass

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()

    def append(self, *a):
        pass

    def extend(self, *a):
        pass

    def update(self, *a, **k):
        pass


def _inert_class(module, name):
    return type(name, (_Inert,), {"__module__": module})


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    if isinstance(storage, _Inert):
        return _Inert()
    t = torch.empty(0, dtype=storage.dtype)
    return t.set_(storage.untyped_storage(), storage_offset, tuple(size), tuple(stride))


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None):
    return data


_ALLOWED = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("torch._utils", "_rebuild_tensor_v2"): _rebuild_tensor_v2,
    ("torch._utils", "_rebuild_parameter"): _rebuild_parameter,
    ("torch", "Size"): torch.Size,
}


class _StateDictUnpickler(pickle.Unpickler):
    def __init__(self, file, archive, prefix):
        super().__init__(file)
        self._archive, self._prefix, self._storages = archive, prefix, {}

    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return _ALLOWED[(module, name)]
        if module == "torch" and name in _DTYPES:
            return _StorageType(name)
        return _inert_class(module, name)

    def persistent_load(self, pid):
        # ('storage', storage type, key, location, numel): torch/serialization.py
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage" and isinstance(pid[1], _StorageType)):
            return _Inert()
        key = str(pid[2])
        if key not in self._storages:
            raw = bytearray(self._archive.read(f"{self._prefix}/data/{key}"))
            dtype = pid[1].dtype
            self._storages[key] = (torch.frombuffer(raw, dtype=dtype) if len(raw) else torch.empty(0, dtype=dtype))
        return self._storages[key]


def load_state_dict_from_checkpoint(path):
    """``state_dict`` (an OrderedDict of CPU tensors) of a ``.ckpt`` / ``.pt`` file: the safe
    ``torch.load(weights_only=True)`` when the file allows it, the inert-placeholder reader above
    otherwise.  A file that is a bare state dict is returned as is."""
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        obj = None
    if obj is None:
        with zipfile.ZipFile(path) as z:
            pkl = [n for n in z.namelist() if n.endswith("/data.pkl")]
            if len(pkl) != 1:
                raise ValueError(f"{path}: not a torch zip checkpoint")
            prefix = pkl[0][: -len("/data.pkl")]
            byteorder = f"{prefix}/byteorder"
            if byteorder in z.namelist() and z.read(byteorder).decode().strip() != "little":
                raise NotImplementedError("big-endian checkpoints")
            with z.open(pkl[0]) as f:
                obj = _StateDictUnpickler(f, z, prefix).load()
    sd = obj.get("state_dict", obj) if isinstance(obj, dict) else None
    if not isinstance(sd, dict) or not sd or not all(torch.is_tensor(v) for v in sd.values()):
        raise ValueError(f"{path}: no state_dict of tensors found")
    return collections.OrderedDict((k, v.clone()) for k, v in sd.items())
