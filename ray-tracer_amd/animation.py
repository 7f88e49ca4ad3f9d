"""Animation driver for BASELINE config 5: a scene description whose object
coordinates are expressions of the frame number `t`, re-evaluated, re-built (sahBVH on the
host) and re-uploaded every frame.

The reference's own driver for this, src/SceneParser.hs, is dead code that no longer
compiles against src/Types.hs, so there is NO working reference behaviour to
match here; what is restated is the documented intent:
  * sections CONFIG / LIGHTS / COLORS / MATERIAL / SHAPES (or OBJECTS) ... END
    (SceneParser.hs:60-73, sample file `sdl`),
  * the expression language of SceneParser.hs:291-395
        Expression := Expression [+-] Term | Term
        Term       := Term [*/] Factor | [- sin cos] Factor | Factor
        Factor     := ( Expression ) | Float | t
    evaluated in single precision (evalExpr, :380-384),
  * Sphere name = {x y z} radius material ; Triangle name = {a} {b} {c} material.
The sample's 8-field `Material` (ambient diffuse specular phong reflection-colour
refraction-index attenuation-colour gloss, SceneParser.hs:196-209) predates the current
6-field one (src/Types.hs:27-37); OUR mapping: diffuse, specular, phong as given,
reflection index = mean of the reflection colour, refraction index as given, attenuation
colour through makeMaterial's log (src/Surfaces.hs:22-24); ambient and gloss are dropped
(the world ambient comes from CONFIG AMBIENT; glossiness is 1/phong in the current tracer).

Host-side Python: none of this is on the trace/shade path.
"""
from __future__ import annotations

import ctypes as C
import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

from . import (Config, Light, Material, Scene, Sphere, REF_TRIANGLE, config_to_world, f32,
               make_triangle, RtError)

F = np.float32


# ------------------------------------------------------------------ expressions
class Expr:
    """Parse once, evaluate per frame.  Arithmetic in float32 like Haskell's Float."""

    _tok = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+)|(sin|cos|t)\b|([-+*/()]))")

    def __init__(self, text: str):
        self.text = text
        self.toks: List[Tuple[str, str]] = []
        pos = 0
        text = text.strip()
        while pos < len(text):
            m = self._tok.match(text, pos)
            if not m:
                raise RtError(f"bad expression {text!r} at {pos}")
            if m.group(1):
                self.toks.append(("num", m.group(1)))
            elif m.group(2):
                self.toks.append(("id", m.group(2)))
            else:
                self.toks.append(("op", m.group(3)))
            pos = m.end()
        self.i = 0
        self.tree = self._expression()
        if self.i != len(self.toks):
            raise RtError(f"trailing input in expression {text!r}")

    def _peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else (None, None)

    def _expression(self):
        node = self._term()
        while self._peek() in (("op", "+"), ("op", "-")):
            op = self.toks[self.i][1]
            self.i += 1
            node = (op, node, self._term())
        return node

    def _term(self):
        node = self._unary()
        while self._peek() in (("op", "*"), ("op", "/")):
            op = self.toks[self.i][1]
            self.i += 1
            node = (op, node, self._factor())
        return node

    def _unary(self):
        k, v = self._peek()
        if (k, v) == ("op", "-"):
            self.i += 1
            return ("neg", self._factor())
        if k == "id" and v in ("sin", "cos"):
            self.i += 1
            return (v, self._factor())
        return self._factor()

    def _factor(self):
        k, v = self._peek()
        if (k, v) == ("op", "("):
            self.i += 1
            node = self._expression()
            if self._peek() != ("op", ")"):
                raise RtError(f"missing ) in {self.text!r}")
            self.i += 1
            return node
        if k == "num":
            self.i += 1
            return ("num", F(v))
        if (k, v) == ("id", "t"):
            self.i += 1
            return ("t",)
        raise RtError(f"unexpected {v!r} in expression {self.text!r}")

    def __call__(self, t: float) -> float:
        return float(self._ev(self.tree, F(t)))

    def _ev(self, n, t):
        op = n[0]
        if op == "num":
            return n[1]
        if op == "t":
            return t
        if op == "neg":
            return F(-self._ev(n[1], t))
        if op == "sin":
            return F(math.sin(float(self._ev(n[1], t))))
        if op == "cos":
            return F(math.cos(float(self._ev(n[1], t))))
        a, b = self._ev(n[1], t), self._ev(n[2], t)
        with np.errstate(all="ignore"):
            return F({"+": a + b, "-": a - b, "*": a * b, "/": a / b}[op])


def _split_point_exprs(body: str) -> List[Expr]:
    """`{e1 e2 e3}`: whitespace separates the three expressions, but not inside parentheses."""
    parts, depth, cur = [], 0, ""
    for ch in body.strip():
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch.isspace() and depth == 0:
            if cur:
                parts.append(cur)
                cur = ""
        else:
            cur += ch
    if cur:
        parts.append(cur)
    if len(parts) != 3:
        raise RtError(f"a point needs three expressions: {{{body}}}")
    return [Expr(p) for p in parts]


# ------------------------------------------------------------------ scene description
@dataclass
class SceneDescription:
    eye: Tuple[float, float, float] = (0, 0, 0)
    look_at: Tuple[float, float, float] = (1, 0, 0)
    up: Tuple[float, float, float] = (0, 1, 0)
    view: Tuple[int, int, int] = (8, 6, 7)
    ambient: Tuple[float, float, float] = (0.1, 0.1, 0.1)
    lights: List[Light] = field(default_factory=list)
    materials: List[Material] = field(default_factory=list)
    material_index: Dict[str, int] = field(default_factory=dict)
    shapes: List[tuple] = field(default_factory=list)  # ("sphere", [3 Expr], Expr, mat) | ("triangle", [[3 Expr]]*3, mat)

    def scene_at(self, t: float) -> Scene:
        """evalObjectExpr (SceneParser.hs:392-395) for every shape, then sahBVH."""
        spheres, tris, order = [], [], []
        for sh in self.shapes:
            if sh[0] == "sphere":
                c = [e(t) for e in sh[1]]
                order.append(len(spheres))
                spheres.append(Sphere((f32 * 3)(*c), sh[2](t), sh[3], 0))
            else:
                pts = [tuple(e(t) for e in p) for p in sh[1]]
                order.append(REF_TRIANGLE | len(tris))
                tris.append(make_triangle(pts[0], pts[1], pts[2], sh[2]))
        return Scene.create(spheres, tris, order, self.materials, self.lights)

    def world_at(self, t: float, config: Config):
        cfg = Config(**{**config.__dict__})
        cfg.eye, cfg.look_at, cfg.up = self.eye, self.look_at, self.up
        cfg.view_width, cfg.view_height, cfg.view_distance = self.view
        w = config_to_world(cfg, self.scene_at(t))
        for i in range(3):
            w.camera.ambient[i] = self.ambient[i]
        return w


def _section(text: str, *names: str) -> str:
    for n in names:
        m = re.search(r"^\s*" + n + r"\s*$(.*?)^\s*END\s*$", text, re.S | re.M)
        if m:
            return m.group(1)
    return ""


def _floats(s: str) -> List[float]:
    return [float(x) for x in s.replace("{", " ").replace("}", " ").split()]


def parse_scene_description(text: str) -> SceneDescription:
    d = SceneDescription()
    colors: Dict[str, Tuple[float, float, float]] = {}

    def color(tokens: List[str]):
        """an identifier, or an anonymous `Color r g b`; returns (colour, tokens consumed)"""
        if tokens[0] == "Color":
            return tuple(float(x) for x in tokens[1:4]), 4
        if tokens[0] not in colors:
            raise RtError(f"unknown colour {tokens[0]!r}")
        return colors[tokens[0]], 1

    for line in _section(text, "COLORS").splitlines():
        m = re.match(r"\s*Color\s+(\w+)\s*=\s*(.*)$", line)
        if m:
            colors[m.group(1)] = tuple(_floats(m.group(2))[:3])
    for line in _section(text, "CONFIG").splitlines():
        tok = line.split(None, 1)
        if len(tok) < 2:
            continue
        k, v = tok
        if k == "EYE":
            d.eye = tuple(_floats(v)[:3])
        elif k == "LOOKAT":
            d.look_at = tuple(_floats(v)[:3])
        elif k == "UP":
            d.up = tuple(_floats(v)[:3])
        elif k == "VIEWPLANE":
            w, h, dist = _floats(v)[:3]
            d.view = (int(math.floor(w)), int(math.floor(h)), int(dist))  # buildSceneConfig :94
        elif k == "AMBIENT":
            d.ambient = color(v.split())[0]
    for line in _section(text, "LIGHTS").splitlines():
        m = re.match(r"\s*Light\s*\{([^}]*)\}\s*(.*)$", line)
        if m:
            c, _ = color(m.group(2).split())
            d.lights.append(Light((f32 * 3)(*_floats(m.group(1))[:3]), (f32 * 3)(0, 0, 0),
                                  (f32 * 3)(0, 0, 0), (f32 * 3)(*c)))
    for line in _section(text, "MATERIAL", "MATERIALS").splitlines():
        m = re.match(r"\s*Material\s+(\w+)\s*=\s*(.*)$", line)
        if not m:
            continue
        tok = m.group(2).split()
        vals, i = [], 0
        for kind in "cccfcfcf":  # amb dif spe bp refl refr attn gloss  (SceneParser.hs:196-209)
            if kind == "c":
                c, n = color(tok[i:])
                vals.append(c)
                i += n
            else:
                vals.append(float(tok[i]))
                i += 1
        _amb, dif, spe, bp, refl, refr, attn, _gloss = vals
        with np.errstate(divide="ignore"):
            la = [float(np.log(F(x))) for x in attn]     # makeMaterial (Surfaces.hs:22-24)
        d.material_index[m.group(1)] = len(d.materials)
        d.materials.append(Material((f32 * 3)(*dif), (f32 * 3)(*spe), bp, float(F(sum(refl) / 3.0)),
                                    refr, (f32 * 3)(*la)))
    for line in _section(text, "SHAPES", "OBJECTS").splitlines():
        m = re.match(r"\s*(Sphere|Triangle)\s+(\w+)\s*=\s*(.*)$", line)
        if not m:
            continue
        pts = re.findall(r"\{([^}]*)\}", m.group(3))
        rest = m.group(3)[m.group(3).rfind("}") + 1:].split()
        if rest[-1] not in d.material_index:
            raise RtError(f"unknown material {rest[-1]!r}")
        mat = d.material_index[rest[-1]]
        if m.group(1) == "Sphere":
            d.shapes.append(("sphere", _split_point_exprs(pts[0]), Expr(" ".join(rest[:-1])), mat))
        else:
            d.shapes.append(("triangle", [_split_point_exprs(p) for p in pts[:3]], mat))
    return d


def render_sequence(renderer, description: SceneDescription, config: Config, grids, frames,
                    flags: int = 0, rebuild_every: int = 1):
    """The animation loop: for every frame number, re-evaluate the shapes, rebuild the BVH on
    the host (sahBVH), upload, render.  Yields (t, rgb [H*W,3] float32, Stats).
    rebuild_every = k > 1: only every k-th frame is rebuilt and uploaded; the frames between keep
    that tree's topology and have their boxes refitted on the GPU (rt_refit_async) -- a
    tree-changing mode: same closest hits, ties between coincident surfaces follow the stale
    tree's leaf order."""
    renderer.set_samples(grids)
    for k, t in enumerate(frames):
        if k % max(1, rebuild_every) == 0:
            renderer.set_world(description.world_at(t, config))
        else:
            renderer.refit(description.scene_at(t))
        rgb, _, st = renderer.render(flags)
        yield t, rgb, st


# A description in the dialect above with the orbits of the reference's sample (`sdl`:46-52:
# spheres at 7*(sin, 0, cos)(t*3.14/50) etc. around a glass sphere) -- our own text, used by
# bench.py's workload "c5".
ORBIT_DESCRIPTION = """
CONFIG
  EYE {15 2 15}
  LOOKAT {-1 -1 -1}
  UP {0 1 0}
  VIEWPLANE {16 9 8}
  AMBIENT Color 0.1 0.1 0.1
END
LIGHTS
  Light {50 20 0} Color 0.5 0.5 0.5
  Light {3 2 20} Color 0.2 0.2 0.2
END
COLORS
  Color purple = 0.5 0.2 0.5
  Color blue = 0.2343 0.625 0.8125
  Color white = 1.0 1.0 1.0
  Color black = 0.0 0.0 0.0
  Color grey = 0.6 0.6 0.6
  Color red = 1.0 0.0 0.0
  Color yellow = 1 0.8 0.4
  Color emerald = 0.2 1 0.1
  Color glass = 0.1 0.2 0.4
END
MATERIAL
  Material mat_glass = glass glass grey 10.0 white 1.2 white 0.0
  Material mat_red_refl = red red grey 1000.0 grey 0.0 black 0.0
  Material mat_emerald = emerald emerald grey 100.0 grey 0.0 black 0.0
  Material mat_blue = blue blue grey 1000.0 black 0.0 black 0.0
  Material mat_floor = yellow yellow white 10.0 black 0.0 black 0.0
END
SHAPES
  Sphere s1 = {0 0 0} 3 mat_glass
  Sphere s2 = {sin(t*3.14/50)*7 0 cos(t*3.14/50)*7} 1 mat_red_refl
  Sphere s3 = {0 sin(t*3.14/50)*8 cos(t*3.14/50)*8} 1 mat_emerald
  Sphere s4 = {sin(t*3.14/50)*6 cos(t*3.14/50)*6 0} 1 mat_blue
  Triangle t1 = {-30 -9 -30} {-30 -9 30} {30 -9 30} mat_floor
  Triangle t2 = {-30 -9 -30} {30 -9 30} {30 -9 -30} mat_floor
END
"""
