class AttnProcessor2_0: pass
class LoRAAttnProcessor2_0: pass
class LoRAXFormersAttnProcessor: pass
class XFormersAttnProcessor: pass
