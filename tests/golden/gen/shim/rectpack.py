"""Stand-in for the absent `rectpack` (reference parsers/texture_packing.py:71-78): newPacker / add_bin / add_rect /
pack / packer[0] with .rid .x .y .width .height, implemented as a shelf packer.  Texture lookups are relative to a
texture's own offset and never leave its rectangle, so the layout does not influence any rendered value."""


class _Rect:
    def __init__(self, x, y, w, h, rid):
        self.x, self.y, self.width, self.height, self.rid = x, y, w, h, rid


class _Packer:
    def __init__(self):
        self.bins, self.rects, self.packed = [], [], [[]]

    def add_bin(self, w, h):
        self.bins.append((w, h))

    def add_rect(self, w, h, rid=None):
        self.rects.append((w, h, rid))

    def pack(self):
        bw, bh = self.bins[0]
        x = y = shelf = 0
        out = []
        for w, h, rid in sorted(self.rects, key=lambda r: (-r[1], -r[0])):
            if w > bw:
                continue
            if x + w > bw:
                x, y, shelf = 0, y + shelf, 0
            if y + h > bh:
                continue
            out.append(_Rect(x, y, w, h, rid)); x += w; shelf = max(shelf, h)
        self.packed = [out]

    def __getitem__(self, k):
        return self.packed[k]


class PackerBBF(_Packer): pass
class PackerBNF(_Packer): pass
class PackerBFF(_Packer): pass


def newPacker(*a, **k):
    return _Packer()


float2dec = None
