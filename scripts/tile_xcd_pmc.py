"""12 eager frames at N = 200 with srukf_debug_set "tile_xcd" = argv[1] (for rocprofv3 --pmc FETCH_SIZE: traffic of k_gmw_persist per launch)."""
import sys
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package(); synth, srukf = pkg.synth, pkg.srukf
srukf.debug_set_global("tile_xcd", int(sys.argv[1])); srukf.debug_set_global("graphs", 0)
N = 200; p = synth.scene_params(); sc = synth.make_scene(N, 20, seed=0, p=p)
f = srukf.Filter(N, p); f.set_state(sc["X0"], sc["S0"]); f.stage_sequence(sc["odo"], sc["z"], sc["matched"])
f.run_frames_async(0, 12); f.synchronize()
