import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as ra
from raytrace_amd import scenes
sc = scenes.lucy_class(28055742, 2160, 3840)
hs = ra.HostScene(sc)
print("lucy", hs.desc.contents.tree_depth, hs.desc.contents.n_nodes, flush=True)
