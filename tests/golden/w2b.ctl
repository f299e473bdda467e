# secondary control file (second command-line argument of G-PhoCS, GPhoCS.c:35-43, 154-164):
# GENERAL-INFO keys given here override the primary file's, the MIG-BANDS module REPLACES its band set
GENERAL-INFO-START
	trace-file          w2.trace
	random-seed         4711
	mcmc-iterations	  50
	iterations-per-log  25
	finetune-theta		0.06
	finetune-mixing		0.002
	mig-rate-print		0.01
	mig-rate-beta		0.0000000500
GENERAL-INFO-END

MIG-BANDS-START
	BAND-START
       source  B
       target  A
       mig-rate-alpha 0.003
	BAND-END

	BAND-START
       source  C
       target  AB
	BAND-END

	BAND-START
       source  A
       target  C
       mig-rate-print 0.1
	BAND-END
MIG-BANDS-END
