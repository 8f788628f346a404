class UNet2DConditionLoadersMixin: pass
class FromSingleFileMixin: pass
class LoraLoaderMixin: pass
class TextualInversionLoaderMixin: pass
