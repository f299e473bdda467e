GENERAL-INFO-START

	seq-file            q6.seq
	trace-file          q6.trace
	locus-mut-rate          CONST
	num-loci            2
	random-seed         12345
	mcmc-iterations	  8
	iterations-per-log  4
	logs-per-line       10

	find-finetunes		FALSE
	finetune-coal-time	0.01		
	finetune-mig-time	0.3		
	finetune-theta		0.04
	finetune-mig-rate	0.02
	finetune-tau		0.0000008
	finetune-mixing		0.003

	tau-theta-print		10000.0
	tau-theta-alpha		1.0
	tau-theta-beta		10000.0

	mig-rate-print		0.001
	mig-rate-alpha		0.002
	mig-rate-beta		0.0000001000

GENERAL-INFO-END

CURRENT-POPS-START	

	POP-START
		name		A
		samples		s0 d s1 d s2 d s3 d s4 d s5 d
	POP-END

	POP-START
		name		B
		samples		s6 d s7 d s8 d s9 d s10 d s11 d
	POP-END

	POP-START
		name		C
		samples		s12 d s13 d s14 d s15 d s16 d s17 d
	POP-END

	POP-START
		name		D
		samples		s18 d s19 d s20 d s21 d s22 d s23 d
	POP-END

	POP-START
		name		E
		samples		s24 d s25 d s26 d s27 d s28 d s29 d
	POP-END

	POP-START
		name		F
		samples		s30 d s31 d s32 d s33 d s34 d s35 d
	POP-END

CURRENT-POPS-END

ANCESTRAL-POPS-START

	POP-START
		name			AB
		children		A		B
		tau-initial	0.000005000
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABC
		children		AB		C
		tau-initial	0.000008000
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCD
		children		ABC		D
		tau-initial	0.000012800
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCDE
		children		ABCD		E
		tau-initial	0.000020480
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			root
		children		ABCDE		F
		tau-initial	0.000102400
		tau-beta		20000.0	
		finetune-tau			0.00000286
	POP-END

ANCESTRAL-POPS-END

MIG-BANDS-START	
	BAND-START		
       source  A
       target  B
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  B
       target  A
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  C
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  E
       target  D
       mig-rate-print 0.1
	BAND-END

MIG-BANDS-END
