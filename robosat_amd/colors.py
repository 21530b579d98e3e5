"""Colour palettes for masks and probability PNGs (same names and values as reference ``robosat/colors.py``)."""

import colorsys

# Mapbox-themed colours, https://www.mapbox.com/base/styling/color/
MAPBOX = {
    "dark": "#404040", "gray": "#eeeeee", "light": "#f8f8f8", "white": "#ffffff", "cyan": "#3bb2d0", "blue": "#3887be",
    "bluedark": "#223b53", "denim": "#50667f", "navy": "#28353d", "navydark": "#222b30", "purple": "#8a8acb",
    "teal": "#41afa5", "green": "#56b881", "yellow": "#f1f075", "mustard": "#fbb03b", "orange": "#f9886c",
    "red": "#e55e5e", "pink": "#ed6498",
}


def rgb(name):
    h = MAPBOX[name]
    return int(h[1:3], 16), int(h[3:5], 16), int(h[5:7], 16)


def make_palette(*colors):
    """Flat PIL palette ``[r0, g0, b0, r1, ...]`` from colour names."""

    return [v for c in colors for v in rgb(c)]


def color_string_to_rgb(color):
    return [int(v) for v in color.split(",")]


def continuous_palette_for_color(color, bins=256):
    """``bins`` shades of one colour: its hue and value with saturation (i+1)/bins (reference colors.py:70-95)."""

    r, g, b = (v / 255 for v in rgb(color))
    h, _, v = colorsys.rgb_to_hsv(r, g, b)
    palette = []
    for i in range(bins):
        palette.extend(int(c * 255) for c in colorsys.hsv_to_rgb(h, (1 / bins) * (i + 1), v))
    return palette
