"""Inert stand-in for tensordict==0.5: a dict with a batch_size attribute."""


class TensorDictBase(dict):
    pass


class TensorDict(TensorDictBase):
    def __init__(self, source=None, batch_size=None, **kwargs):
        super().__init__(source or {})
        self.batch_size = batch_size
