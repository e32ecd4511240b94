"""Regenerates tests/golden/c1_32_radiance.npy with the CPU oracle (Cornell C1, 32x32, 4 accumulated samples)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from rtxpt_amd import scenes
from oracle import ptref
sc, cam = scenes.cornell_box("C1")
o = ptref.Oracle(); o.set_scene(sc); o.set_camera(scenes.bridge_camera(32, 32, **cam)); o.set_settings(scenes.config_settings("C1")); o.resize(32, 32); o.render(0, 4)
np.save(os.path.join(os.path.dirname(__file__), "c1_32_radiance.npy"), o.radiance())
