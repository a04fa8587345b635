"""Attribute dictionary + ``save_hyperparameters`` used by the sampler classes.

The reference's classes are LightningModules whose ``save_hyperparameters()`` records the
constructor arguments in ``self.hparams``; this is the same behaviour without Lightning.
"""
import inspect


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class HyperparametersMixin:
    """``save_hyperparameters()`` copies the calling ``__init__``'s arguments (including
    ``**kwargs``) into ``self.hparams``; repeated calls from subclasses extend it."""

    @property
    def hparams(self):
        if "_hparams" not in self.__dict__:
            self.__dict__["_hparams"] = AttrDict()
        return self.__dict__["_hparams"]

    def save_hyperparameters(self):
        frame = inspect.currentframe().f_back
        info = inspect.getargvalues(frame)
        for name in info.args:
            if name != "self":
                self.hparams[name] = info.locals[name]
        if info.keywords:
            for k, v in info.locals[info.keywords].items():
                self.hparams[k] = v
