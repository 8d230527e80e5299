exec(open('/tmp/analyse_mfma.py').read().split("# layout check")[0])
lay=np.zeros((T,16,16)); Chw=np.zeros((T,16,16))
for l in range(64):
    g=l//16; j=l%16
    for r in range(4):
        lay[:,4*r+g,j]=D[:,l,r]
        Chw[:,4*r+g,j]=C[:,4*g+r,j]
ref=np.einsum('tik,tkj->tij',A,B)+Chw
print("layout max rel err", np.max(np.abs(lay-ref)/(np.abs(ref)+1e-300)))
exec("def rd(x): return float(x)\ndef fma(a,b,c): return rd(Fr(a)*Fr(b)+Fr(c))")
exec(open('/tmp/analyse_mfma.py').read().split("def fma(a,b,c): return rd(Fr(a)*Fr(b)+Fr(c))")[1].split("score={}")[0])
score={}; n=0
for t in range(T):
    for i in range(16):
        for j in range(0,16,3):
            got=lay[t,i,j]; cs=cands(A[t,i],B[t,:,j],Chw[t,i,j]); n+=1
            for k,v in cs.items(): score[k]=score.get(k,0)+(v==got)
print(n, sorted(score.items(), key=lambda kv:-kv[1])[:6])
