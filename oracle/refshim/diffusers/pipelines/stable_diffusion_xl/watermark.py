class StableDiffusionXLWatermarker:
    def apply_watermark(self, images):
        return images
