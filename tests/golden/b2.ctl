GENERAL-INFO-START

	seq-file            b2.seq
	trace-file          b2.trace
	locus-mut-rate          CONST
	num-loci            10
	random-seed         12345
	mcmc-iterations	  24
	iterations-per-log  8
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
	mig-rate-beta		0.0000000400

GENERAL-INFO-END

CURRENT-POPS-START	

	POP-START
		name		A
		samples		s0 d
	POP-END

	POP-START
		name		B
		samples		s1 d
	POP-END

	POP-START
		name		C
		samples		s2 d
	POP-END

	POP-START
		name		D
		samples		s3 d
	POP-END

	POP-START
		name		E
		samples		s4 d
	POP-END

	POP-START
		name		F
		samples		s5 d
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
       target  C
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  C
       target  D
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  E
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  E
       target  F
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  B
       target  A
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  C
       target  B
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

	BAND-START		
       source  F
       target  E
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  A
       target  C
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  B
       target  D
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  C
       target  E
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  F
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  C
       target  A
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  B
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  E
       target  C
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  F
       target  D
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  A
       target  D
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  A
       mig-rate-print 0.1
	BAND-END

MIG-BANDS-END
