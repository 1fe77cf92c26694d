import json,sys
d=json.load(open(sys.argv[1]))
print("value", d["value"], "fps", d["frames_per_s"], "ms", d["ms_per_step"])
for k in ("projection","jacobi","jacobi_diffuse_velocity","host_enqueue_us_per_frame"):
    print(k, d.get(k))
