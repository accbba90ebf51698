import json, sys
import numpy as np
a = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
for i in sorted(set(a["restarts"]) & set(b["restarts"]), key=int):
    x, y = a["restarts"][i], b["restarts"][i]
    print("restart", i, "nconv", x["nconv"], y["nconv"], "max |dRitz| / |Ritz|max = %.3e" % (np.abs(np.array(x["re"]) - np.array(y["re"])).max() / np.abs(np.array(y["re"])).max()))
