"""import-time stub for the absent `opencv-python` (only touched when a scene declares textures)."""
def imread(*a, **k): raise NotImplementedError("cv2 is not available")
