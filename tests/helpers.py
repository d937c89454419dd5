"""Shared helpers for the tests: fixture loading."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENE_FIXTURES = ["tiny0", "tiny5", "cornell", "bunny2k", "sponza12k", "texroom", "deeptree"]


def load_fixture(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def scene_from_fixture(fx):
    """Rebuild the scene dict (meshes, spheres) a golden fixture was generated from."""
    meshes = []
    for i in range(int(fx["n_meshes"])):
        m = {"verts": fx[f"m{i}_verts"], "lig": fx[f"m{i}_lig"], "col": fx[f"m{i}_col"], "type": int(fx[f"m{i}_type"])}
        if f"m{i}_tri_cols" in fx:
            m["tri_cols"] = fx[f"m{i}_tri_cols"]
        if f"m{i}_tri_tex" in fx:
            m["tri_tex"], m["tri_ts"] = fx[f"m{i}_tri_tex"], fx[f"m{i}_tri_ts"]
        meshes.append(m)
    spheres = [{"rad": float(r), "pos": p, "lig": l, "col": c, "type": int(t)}
               for r, p, l, c, t in zip(fx["sph_rad"], fx["sph_pos"], fx["sph_lig"], fx["sph_col"], fx["sph_type"])]
    textures = [fx[f"tex{k}"] for k in range(int(fx["n_textures"]))] if "n_textures" in fx else []
    return {"meshes": meshes, "spheres": spheres, "textures": textures}
