import sys, numpy as np, ctypes as C
sys.path.insert(0,'.')
from hipstr_amd import capi
hmm = capi.load_hmm(); ora = capi.load_oracle()
def one_locus(sb, l):
    """extract locus l of a synth batch into a 1-locus/1-read Batch"""
    b = sb.ptr.contents
    opt_base = int(np.sum(np.ctypeslib.as_array(b.blk_nopts, shape=(3*sb.n_loci,))[:3*l]))
    nopts = np.ctypeslib.as_array(b.blk_nopts, shape=(3*sb.n_loci,))[3*l:3*l+3]
    opt_off = np.ctypeslib.as_array(b.opt_off, shape=(opt_base+int(nopts.sum())+1,))
    seq = C.string_at(b.seq, int(opt_off[-1]))
    blocks=[]; c=opt_base
    for k in range(3):
        opts=[seq[opt_off[c+o]:opt_off[c+o+1]].decode() for o in range(nopts[k])]; c+=nopts[k]
        blocks.append((int(b.blk_start[3*l+k]), int(b.blk_end[3*l+k]), opts))
    r = int(b.read_off[l]); bo0,bo1=int(b.base_off[r]),int(b.base_off[r+1])
    rd = dict(seq=C.string_at(b.bases,bo1)[bo0:bo1].decode(), qual=C.string_at(b.quals,bo1)[bo0:bo1].decode(), start=int(b.read_start[r]),
              cigar=[(chr(b.cigar_op[i][0]) if isinstance(b.cigar_op[i],bytes) else chr(b.cigar_op[i]), int(b.cigar_len[i])) for i in range(int(b.cigar_off[r]), int(b.cigar_off[r+1]))])
    nb = capi.Batch(); A = nb.add_locus(blocks, int(b.period[l]), [b.stutter[6*l+i] for i in range(6)], [rd]); nb.finalize()
    return nb, blocks, A
sb = capi.SynthBatch(n_loci=2, reads_per_locus=50, n_str_alleles=4, seed=1)
for l in range(2):
    nb, blocks, A = one_locus(sb, l)
    F0=len(blocks[0][2][0]); F2=len(blocks[2][2][0])
    print("locus",l,"lf..",blocks[0][2][0][-8:],"STR",[o[:10] for o in blocks[1][2]],"rf",blocks[2][2][0][:8])
    for k in range(A):
        hf=np.zeros(1024,np.int32); hr=np.zeros(1024,np.int32)
        assert ora.oracle_debug_row_h(nb.ptr,k,hf.ctypes.data_as(capi._i32p),hr.ctypes.data_as(capi._i32p),1024)==0
        opts=np.zeros(3,np.int32); ora.oracle_allele_options(np.array([1,len(blocks[1][2]),1],np.int32).ctypes.data_as(capi._i32p),k,opts.ctypes.data_as(capi._i32p))
        B=len(blocks[1][2][opts[1]])
        for side,(hh,Fl,Ft) in enumerate(((hf,F0,F2),(hr,F2,F0))):
            for which in (0,1):
                rows=np.zeros(1024,np.uint32); n=hmm.hipstr_debug_rows(nb.ptr,k,side,which,rows.ctypes.data_as(C.POINTER(C.c_uint32)),1024)
                mine=((rows[:n]>>8)&15).astype(int)
                want = hh[0:Fl] if which==0 else hh[Fl+B:Fl+B+Ft]
                bad=[i for i in range(1,n) if mine[i]!=want[i]]
                if bad: print("  k",k,"side",side,"which",which,"bad rows",bad,"mine",mine[bad],"want",want[bad])
print("---- detail locus 1")
nb, blocks, A = one_locus(sb, 1)
print("LF", blocks[0][2][0]); print("STR0", blocks[1][2][0]); print("RF", blocks[2][2][0])
hf=np.zeros(1024,np.int32); hr=np.zeros(1024,np.int32)
ora.oracle_debug_row_h(nb.ptr,0,hf.ctypes.data_as(capi._i32p),hr.ctypes.data_as(capi._i32p),1024)
rows=np.zeros(1024,np.uint32); n=hmm.hipstr_debug_rows(nb.ptr,0,0,0,rows.ctypes.data_as(C.POINTER(C.c_uint32)),1024)
print("mine", ((rows[:n]>>8)&15).tolist()); print("want", hf[:n].tolist())
