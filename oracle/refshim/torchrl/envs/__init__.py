class EnvBase:
    def __init__(self, device="cpu", **kwargs):
        self.device = device

    def to(self, device):
        self.device = device
        return self
