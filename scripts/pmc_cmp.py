import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline_all"]["shade"]
    print(f, d["kernel_ms"]["shade"], {k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ("mfma_busy_frac","mfma_valu_coexec_frac_of_busy","valu_insts_per_32_entry_tile","clock_GHz","wave_cycles_frac","traffic")})
