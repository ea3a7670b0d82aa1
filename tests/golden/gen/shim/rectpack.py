"""import-time stub for the absent `rectpack` (texture atlas packing is outside the pt scope)."""
class PackerBBF: pass
class PackerBNF: pass
class PackerBFF: pass
def newPacker(*a, **k): raise NotImplementedError("rectpack is not available")
float2dec = None
