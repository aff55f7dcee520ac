for rep in 1 2; do for lk in 0 192 256; do
  HOLO_CONV_BF16T_LONGK=$lk bash scripts/gpu_ops128.sh ab_${lk}_$rep bf16 > /dev/null 2>&1
  python3 - <<PY
import ast, json
ops=[ast.literal_eval(l) for l in open("gpurun_out/ab_${lk}_$rep/ops.txt")]
small=sum(o["ms"] for o in ops if o["op"]=="conv" and o.get("ksz")==3 and o.get("stride")==1 and o["out_dim"]<=32)
big=sum(o["ms"] for o in ops if o["op"]=="conv" and o.get("ksz")==3 and o.get("stride")==1 and o["out_dim"]>32)
d=json.load(open("gpurun_out/ab_${lk}_$rep/bench.json"))
print("longk $lk rep $rep: small-level 3x3x3 convs %.3f ms, 64^3+128^3 %.3f ms, step %.3f ms" % (small, big, d["ms_per_step"]))
PY
done; done
