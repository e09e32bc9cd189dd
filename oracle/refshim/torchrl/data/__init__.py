class _Spec:
    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs
        self.shape = kwargs.get("shape")


class TensorSpec(_Spec): pass
class BoundedTensorSpec(_Spec): pass
class UnboundedContinuousTensorSpec(_Spec): pass
class UnboundedDiscreteTensorSpec(_Spec): pass
class CompositeSpec(_Spec): pass
