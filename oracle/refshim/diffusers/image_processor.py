class VaeImageProcessor:
    def __init__(self, *a, **k): pass
