"""`lib_yolo/data.py:80-86` -- only `Prior` is on the inference path (anchor h/w as image fractions)."""


class Prior:
    def __init__(self, h, w):
        self.h = h
        self.w = w

    def __repr__(self):
        return '<Prior - h: {}, w: {}>'.format(self.h, self.w)
