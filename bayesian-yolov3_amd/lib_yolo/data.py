"""`lib_yolo/data.py:80-95`: `Prior` (anchor h/w as image fractions) and `DetLayerInfo` (what `tfdata.encode_boxes` needs to
know of a detection layer).  The reference's numpy ground-truth helpers of this file (`calc_gt` ...) are not used by its
training graph (that is `lib_yolo/tfdata.py`) and have no counterpart here."""


class Prior:
    def __init__(self, h, w):
        self.h = h
        self.w = w

    def __repr__(self):
        return '<Prior - h: {}, w: {}>'.format(self.h, self.w)


class DetLayerInfo:
    def __init__(self, h, w, priors):
        self.h = h
        self.w = w
        self.priors = priors

    def __repr__(self):
        return '<DetLayerInfo - h: {}, w: {}, priors: {}>'.format(self.h, self.w, self.priors)
